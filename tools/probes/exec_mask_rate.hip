// Does gfx950 skip the passes of a wave64 VALU instruction whose lanes are all masked off?  (If it did, the 20-lane vector
// phases of the solver would cost half when run under EXEC = lanes 0..31.)  Times a long stream of independent v_fma_f64 /
// v_mov_b32_dpp under EXEC = 64, 32 and 16 active lanes.   hipcc --offload-arch=gfx950 -O3 exec_mask_rate.hip -o exec_mask_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, int active) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
    const double m = 1.0000001, c = 1e-9;
    if ((int)(threadIdx.x & 63) < active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], m, c);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_dpp(int* out, int iters, int active) {
    int a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    if ((int)(threadIdx.x & 63) < active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_update_dpp(0, a[i], 0x111, 0xf, 0xf, true) + 1;
        }
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d; int* di;
    const int blocks = 256 * 4, threads = 256, iters = 20000;
    hipMalloc(&d, sizeof(double) * blocks * threads); hipMalloc(&di, sizeof(int) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind)
        for (int active : {64, 32, 16, 20}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, d, iters, active);
                else hipLaunchKernelGGL(k_dpp, dim3(blocks), dim3(threads), 0, 0, di, iters, active);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            // instructions per SIMD: blocks*4 waves / 1024 SIMDs * iters * 32
            const double inst = (double)blocks * 4 / 1024 * iters * 32 * (kind == 0 ? 1 : 2);
            printf("%s active %2d: %8.3f ms  -> %.2f cycles per wave-instruction at 2.4 GHz\n", kind == 0 ? "v_fma_f64     " : "v_mov_dpp+add ", active,
                   best, best * 1e-3 * 2.4e9 / inst);
        }
    return 0;
}
