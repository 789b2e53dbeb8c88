// Probe: operand / result layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 and the two-MFMA row all-reduce.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    const double x = 1.0 + l;            // distinct per lane
    const double ones = 1.0;
    // (1) D1 = X_A * ones
    double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, ones, 0.0, 0, 0, 0);
    // (2) D2 = ones * D1_B
    double d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(ones, d1, 0.0, 0, 0, 0);
    // (3) transpose probe: X_A * I  with I as B operand: B[k][j] at lane (4k + j)? -> identity = (k == j)
    const int e = l & 15;
    const double idB = ((e >> 2) == (e & 3)) ? 1.0 : 0.0;
    double t = __builtin_amdgcn_mfma_f64_4x4x4f64(x, idB, 0.0, 0, 0, 0);
    // (4) left-multiply probe: E * X_B with E = unit matrix having a single 1 at [0][1] as A operand: A[i][k] at lane (i + 4k)
    const double eA = (e == 0 + 4 * 1) ? 1.0 : 0.0;
    double lm = __builtin_amdgcn_mfma_f64_4x4x4f64(eA, x, 0.0, 0, 0, 0);
    out[l] = d1; out[64 + l] = d2; out[128 + l] = t; out[192 + l] = lm;
}
int main() {
    double* d; hipMalloc(&d, 256 * sizeof(double));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"D1 = X_A*ones", "D2 = ones*D1", "T = X_A*I", "E01*X_B"};
    for (int r = 0; r < 4; ++r) { printf("%s:\n", names[r]); for (int l = 0; l < 32; ++l) printf("%6.0f%s", h[64 * r + l], (l % 16 == 15) ? "\n" : " "); }
    return 0;
}
