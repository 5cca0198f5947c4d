// 16 x 16 diagonal-tile Cholesky shared by the workgroup-per-row kernels (wmf_direct.hip, wmf_wide.hip).
#pragma once
#include "wmf_common.h"

__device__ __forceinline__ float dreadlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// wave 0: Cholesky of the published 16 x 16 diagonal tile and the inverse of its factor -> T.
// Not inlined: one copy serves every block row of every instantiation (keeps the code in the I-cache).
static __device__ __noinline__ bool direct_diag(const float* __restrict__ Dblk, float* __restrict__ T, int lane) {
    float a[16];
    const int row = (lane < 16) ? lane : 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(Dblk + row * 20 + 4 * c);
        a[4 * c] = v.x; a[4 * c + 1] = v.y; a[4 * c + 2] = v.z; a[4 * c + 3] = v.w;
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float dk = dreadlane(a[k], k);
        if (!(dk > 1e-20f)) ok = false;
        const float inv = __builtin_amdgcn_rsqf(dk);
        a[k] *= inv;                                               // lane k: sqrt(dk); lanes > k: L[i][k]
#pragma unroll
        for (int j = k + 1; j < 16; ++j) a[j] -= a[k] * dreadlane(a[k], j);
    }
    // inverse: lane j builds column j of X = L^-1 (entries above the diagonal come out as 0)
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float s = (i == lane) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= dreadlane(a[k], i) * x[k];
        x[i] = s * __builtin_amdgcn_rcpf(dreadlane(a[i], i));
    }
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) T[i * 20 + lane] = x[i];      // T[i][j] = X[i][j]
    }
    return ok;
}

