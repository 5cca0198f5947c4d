// Probe: wmf_split4 / wmf_split4_scaled (wmf_common.h) against the plain C++ split on random and special values.
// hipcc --offload-arch=gfx950 -O3 -std=c++20 -I recmodel_amd/csrc tools/lab/split_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include "wmf_common.h"
__global__ void probe(const float* x, const float* s, unsigned* out_a, unsigned* out_b, unsigned* out_c, int n4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float a = x[4 * i], b = x[4 * i + 1], c = x[4 * i + 2], d = x[4 * i + 3];
    const wmf_u32x4 r = wmf_split4(a, b, c, d);
    for (int k = 0; k < 4; ++k) out_a[4 * i + k] = r[k];
    const wmf_u32x4 q = wmf_split4_scaled(a, b, c, d, s[4 * i], s[4 * i + 1], s[4 * i + 2], s[4 * i + 3]);
    for (int k = 0; k < 4; ++k) out_b[4 * i + k] = q[k];
    // reference
    const float v[4] = {a, b, c, d};
    _Float16 h[4], l[4];
    for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)
        h[k] = (_Float16)v[k];
        l[k] = (_Float16)(v[k] - (float)h[k]);
    }
    unsigned short hb[4], lb[4];
    for (int k = 0; k < 4; ++k) { hb[k] = __builtin_bit_cast(unsigned short, h[k]); lb[k] = __builtin_bit_cast(unsigned short, l[k]); }
    out_c[4 * i] = hb[0] | (hb[1] << 16); out_c[4 * i + 1] = hb[2] | (hb[3] << 16);
    out_c[4 * i + 2] = lb[0] | (lb[1] << 16); out_c[4 * i + 3] = lb[2] | (lb[3] << 16);
}
int main() {
    const int n4 = 1 << 16, n = 4 * n4;
    float* hx = (float*)malloc(n * 4); float* hs = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int e = rand() % 40 - 30;
        hx[i] = ((rand() / (float)RAND_MAX) * 2 - 1) * ldexpf(1.f, e);
        hs[i] = (rand() / (float)RAND_MAX) * 5.f;
    }
    hx[0] = 0.f; hx[1] = -0.f; hx[2] = 1.f; hx[3] = -1.f; hx[4] = 65504.f; hx[5] = 1e-8f; hx[6] = 6e-5f; hx[7] = -3e-7f;
    float *dx, *ds; unsigned *da, *db, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemcpy(ds, hs, n * 4, hipMemcpyHostToDevice);
    probe<<<n4 / 256, 256>>>(dx, ds, da, db, dc, n4);
    unsigned* ha = (unsigned*)malloc(n * 4); unsigned* hc = (unsigned*)malloc(n * 4);
    hipMemcpy(ha, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) if (ha[i] != hc[i]) { if (bad < 10) printf("mismatch word %d: asm %08x ref %08x (x = %g %g)\n", i, ha[i], hc[i], hx[(i / 4) * 4 + 2 * (i % 2)], hx[(i / 4) * 4 + 2 * (i % 2) + 1]); ++bad; }
    printf("wmf_split4 vs reference: %d mismatching words of %d\n", bad, n);
    return 0;
}
