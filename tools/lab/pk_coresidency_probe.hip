// Lab probe (MI355X), second attempt at the round-3 rowsplit finding: a VICTIM workgroup runs the packed {y, b} accumulation
// chain (v_mul_f32 into the high half of a pair, then v_pk_fma_f32 reading the pair) while a DISTURBER workgroup of another kind
// shares its CU -- 256 registers per wave, so exactly two workgroups per CU, one wave of each per SIMD.  The first 256 workgroups
// are victims, the next 256 disturbers (a CU takes its second workgroup after every CU of the XCD has its first).
// Build: hipcc --offload-arch=gfx950 -O2 -o build/pk_probe2 tools/lab/pk_coresidency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, float one, int* sink) {
    extern __shared__ float lds[];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x < 256) {                                       // ---- victim
        const float alo = 1.0f + (float)(gid % 977) * 0.0009765625f, ahi = 2.0f, x = 0.5f;
        float rlo, rhi;
        asm volatile(
            "v_mov_b32 v210, %2\n\t" "v_mov_b32 v211, %3\n\t" "v_mov_b32 v176, %4\n\t" "v_mov_b32 v177, %4\n\t"
            "v_mov_b32 v186, 0\n\t" "v_mov_b32 v187, 0\n\t" "s_mov_b32 s40, %5\n\t" "s_mov_b32 s48, %6\n\t" "s_nop 4\n\t"
            "1:\n\t"
            "v_mul_f32 v211, s48, v211\n\t"
            "v_mul_f32 v209, s48, v211\n\t"
            "v_pk_fma_f32 v[186:187], v[210:211], v[176:177], v[186:187] op_sel:[0,1,0]\n\t"
            "v_mul_f32 v211, s48, v211\n\t"
            "v_pk_fma_f32 v[186:187], v[210:211], v[176:177], v[186:187] op_sel_hi:[1,0,1]\n\t"
            "s_sub_u32 s40, s40, 1\n\t" "s_cmp_lg_u32 s40, 0\n\t" "s_cbranch_scc1 1b\n\t" "s_nop 4\n\t"
            "v_mov_b32 %0, v186\n\t" "v_mov_b32 %1, v187\n\t"
            : "=v"(rlo), "=v"(rhi) : "v"(alo), "v"(ahi), "v"(x), "s"(iters), "s"(one)
            : "v210", "v211", "v209", "v186", "v187", "v176", "v177", "v255", "s40", "s48", "scc");
        out[2 * gid] = rlo; out[2 * gid + 1] = rhi;
        return;
    }
    // ---- disturber: keeps 256 registers allocated (v255 clobber) and hammers one unit
    float a = (float)threadIdx.x, b = 1.5f, c = 0.25f;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    h8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)(0.01f * j); hb[j] = (_Float16)(0.02f * j); }
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    for (int i = 0; i < iters * 2; ++i) {
        if (KIND == 1) {                                          // permlane swaps
            const int xi = __builtin_bit_cast(int, a);
            const auto s = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
            const auto t = __builtin_amdgcn_permlane16_swap((int)s[0], (int)s[1], false, false);
            a = __builtin_bit_cast(float, (int)t[0]) + 1.f;
        } else if (KIND == 2) {                                   // DPP adds
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));
        } else if (KIND == 3) {                                   // readlane / writelane
            int v;
            asm volatile("v_readlane_b32 %0, %1, 5\n\ts_add_u32 %0, %0, 1\n\ts_nop 0\n\tv_writelane_b32 %1, %0, 9" : "=&s"(v), "+v"(a) : : "scc");
        } else if (KIND == 4) {                                   // f16 MFMA
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, acc, 0, 0, 0);
        } else if (KIND == 5) {                                   // ds_bpermute + LDS traffic
            lds[threadIdx.x] = a;
            a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((threadIdx.x * 5) & 63) * 4, __builtin_bit_cast(int, a))) + lds[(threadIdx.x * 3) & 255];
        } else if (KIND == 6) {                                   // transcendental unit
            a = __builtin_amdgcn_rcpf(a + 2.f) + __builtin_amdgcn_sqrtf(b); b += 0.5f;
        } else if (KIND == 7) {                                   // packed f32 of its own
            asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(*(double*)&acc) : "v"(*(double*)&ha));
        } else if (KIND == 8) {                                   // global loads
            a += out[(gid * 17 + i * 4099) & 0xfffff];
        }
    }
    if (a + acc[0] + acc[1] + b == 12345.678f) sink[0] = 1;
}

template <int KIND>
static int run(int iters, const char* what) {
    const int blocks = 512, n = 256 * 256;
    float* d; int* sink;
    (void)hipMalloc(&d, (size_t)(1 << 20) * 4 * 2); (void)hipMalloc(&sink, 4);
    (void)hipMemset(d, 0, (size_t)(1 << 20) * 8);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 1024, 0, d, iters, 1.0f, sink);
    (void)hipDeviceSynchronize();
    std::vector<float> h((size_t)n * 2);
    (void)hipMemcpy(h.data(), d, (size_t)n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d); (void)hipFree(sink);
    long bad_lo = 0, bad_hi = 0;
    for (int g = 0; g < n; ++g) {
        const float alo = 1.0f + (float)(g % 977) * 0.0009765625f;
        float elo = 0.f, ehi = 0.f;
        for (int i = 0; i < 2 * iters; ++i) { elo = fmaf(alo, 0.5f, elo); ehi = fmaf(2.0f, 0.5f, ehi); }
        bad_lo += h[2 * g] != elo; bad_hi += h[2 * g + 1] != ehi;
    }
    printf("disturber %d (%s): victims' wrong low halves %ld, wrong high halves %ld of %d lanes\n", KIND, what, bad_lo, bad_hi, n);
    return bad_lo || bad_hi;
}

int main() {
    int rc = 0;
    rc |= run<0>(20000, "idle");
    rc |= run<1>(20000, "v_permlane32/16_swap");
    rc |= run<2>(20000, "DPP");
    rc |= run<3>(20000, "v_readlane / v_writelane");
    rc |= run<4>(20000, "f16 MFMA");
    rc |= run<5>(20000, "ds_bpermute + LDS");
    rc |= run<6>(20000, "v_rcp / v_sqrt");
    rc |= run<7>(20000, "v_pk_fma_f32");
    rc |= run<8>(20000, "global loads");
    return rc;
}
