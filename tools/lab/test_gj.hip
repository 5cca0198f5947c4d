// standalone check of gj_inv_sweep (wmf_directw.hip) against a host inverse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <utility>
#include "../../recmodel_amd/csrc/wmf_common.h"
__global__ void k(const float* A, float* X) {
    int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    f32x4 a;
    for (int reg = 0; reg < 4; ++reg) a[reg] = A[r * 16 + 4 * q + reg];
    int baddr[4];
    for (int kq = 0; kq < 4; ++kq) baddr[kq] = (r + 16 * kq) * 4;
    bool ok = true;
    gj_inv_sweep(a, baddr, r, q, ok, std::make_integer_sequence<int, 16>{});
    for (int reg = 0; reg < 4; ++reg) X[r * 16 + 4 * q + reg] = a[reg];
}
int main() {
    std::vector<float> A(256), X(256);
    std::vector<double> M(256);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 1.0 : 0.0; for (int k = 0; k < 20; ++k) s += sin(0.37 * (i + 1) * (k + 1)) * sin(0.37 * (j + 1) * (k + 1)) * 0.5; M[i * 16 + j] = s; A[i * 16 + j] = (float)s; }
    float *dA, *dX; hipMalloc(&dA, 1024); hipMalloc(&dX, 1024);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dX);
    hipMemcpy(X.data(), dX, 1024, hipMemcpyDeviceToHost);
    // check A * X = I
    double worst = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += M[i * 16 + k] * X[k * 16 + j]; worst = fmax(worst, fabs(s - (i == j))); }
    printf("max |A X - I| = %g   X[0][0]=%g X[3][7]=%g X[7][3]=%g\n", worst, X[0], X[3 * 16 + 7], X[7 * 16 + 3]);
    return 0;
}
