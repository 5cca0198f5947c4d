// Lab probe (MI355X): does v_pk_fma_f32 read the LOW half of a 64-bit VGPR pair correctly when the instruction in front of it
// wrote the HIGH half, with several waves sharing a SIMD?  Round 3 found solve_rowsplit_kernel's packed {y, b} accumulation
// chain -- v_mul_f32 v211, s48, v211 ; v_pk_fma_f32 .., v[210:211], .. -- flaky in its LOW lanes at two workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O2 -o pk_probe tools/lab/pk_hazard_probe.hip ; run: ./pk_probe [gap]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int GAP>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const float alo = 1.0f + (float)(gid % 977) * 0.0009765625f;         // differs from wave to wave
    const float ahi = 2.0f, x = 0.5f;
    float rlo, rhi;
    asm volatile(
        "v_mov_b32 v210, %2\n\t"
        "v_mov_b32 v211, %3\n\t"
        "v_mov_b32 v176, %4\n\t"
        "v_mov_b32 v177, %4\n\t"
        "v_mov_b32 v186, 0\n\t"
        "v_mov_b32 v187, 0\n\t"
        "s_mov_b32 s40, %5\n\t"
        "s_mov_b32 s41, 1.0\n\t"
        "s_nop 4\n\t"
        "1:\n\t"
        "v_mul_f32 v211, s41, v211\n\t"                            // writes the HIGH half (value unchanged: * 1.0)
        ".rept %6\n\t"
        "s_nop 0\n\t"
        ".endr\n\t"
        "v_pk_fma_f32 v[186:187], v[210:211], v[176:177], v[186:187]\n\t" // lo: += alo * x, hi: += ahi * x
        "v_mul_f32 v211, s41, v211\n\t"
        ".rept %6\n\t"
        "s_nop 0\n\t"
        ".endr\n\t"
        "v_pk_fma_f32 v[186:187], v[210:211], v[176:177], v[186:187] op_sel:[0,1,0]\n\t"
        "s_sub_u32 s40, s40, 1\n\t"
        "s_cmp_lg_u32 s40, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %0, v186\n\t"
        "v_mov_b32 %1, v187\n\t"
        : "=v"(rlo), "=v"(rhi)
        : "v"(alo), "v"(ahi), "v"(x), "s"(iters), "n"(GAP)
        : "v210", "v211", "v186", "v187", "v176", "v177", "v255", "s40", "s41", "scc");
    out[2 * gid] = rlo;
    out[2 * gid + 1] = rhi;
}

template <int GAP>
static int run(int iters) {
    const int blocks = 256 * 2 * 4, n = blocks * 256;            // 256 registers a wave: two waves per SIMD, the second in the upper half of the file
    float* d;
    hipMalloc(&d, (size_t)n * 8);
    hipLaunchKernelGGL(probe<GAP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<float> h((size_t)n * 2);
    hipMemcpy(h.data(), d, (size_t)n * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    long bad_lo = 0, bad_hi = 0;
    for (int g = 0; g < n; ++g) {
        const float alo = 1.0f + (float)(g % 977) * 0.0009765625f;
        float elo = 0.f, ehi = 0.f;
        for (int i = 0; i < 2 * iters; ++i) { elo = fmaf(alo, 0.5f, elo); ehi = fmaf(2.0f, 0.5f, ehi); }
        bad_lo += h[2 * g] != elo;
        bad_hi += h[2 * g + 1] != ehi;
    }
    printf("gap %d: %d lanes, wrong low halves %ld, wrong high halves %ld\n", GAP, n, bad_lo, bad_hi);
    return bad_lo || bad_hi;
}

int main() {
    int rc = 0;
    rc |= run<0>(200);
    rc |= run<1>(200);
    rc |= run<2>(200);
    rc |= run<4>(200);
    return rc;
}
