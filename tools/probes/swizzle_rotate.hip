// Probe (round 4): what ds_swizzle_b32's ROTATE mode does on gfx950 (offset = 0xC000 | dir << 10 | n << 5: rotation inside groups of
// 32 lanes), and what it costs against a DPP row shift.  Build + run: hipcc --offload-arch=gfx950 -O2 swizzle_rotate.hip -o /tmp/swz && /tmp/swz
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PAT> __global__ void map_kernel(int* out) { out[threadIdx.x] = __builtin_amdgcn_ds_swizzle((int)threadIdx.x, PAT); }
// dependent chains of 64-bit moves + adds: DPP row_shr:1 (2 v_mov_b32_dpp + v_add_f64) against ds_swizzle rotate (2 ds_swizzle + v_add_f64)
template <int MODE> __global__ void chain_kernel(double* io, int reps, long long* cyc) {
    double x = io[threadIdx.x];
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int lo = __double2loint(x), hi = __double2hiint(x);
            if (MODE == 0) {
                lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true);
                hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
            } else {
                lo = __builtin_amdgcn_ds_swizzle(lo, 0xC000 | (1 << 10) | (1 << 5));
                hi = __builtin_amdgcn_ds_swizzle(hi, 0xC000 | (1 << 10) | (1 << 5));
            }
            x += __hiloint2double(hi, lo) * 1e-3;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    io[threadIdx.x + blockIdx.x * 64] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    int* d; hipMalloc(&d, 64 * 4); int h[64];
    auto show = [&](const char* name) { hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); printf("%s:", name); for (int i = 0; i < 64; ++i) printf(" %d", h[i]); printf("\n"); };
    hipLaunchKernelGGL((map_kernel<0xC000 | (0 << 10) | (1 << 5)>), 1, 64, 0, 0, d); show("ROTATE dir 0, n 1 (lane i receives)");
    hipLaunchKernelGGL((map_kernel<0xC000 | (1 << 10) | (1 << 5)>), 1, 64, 0, 0, d); show("ROTATE dir 1, n 1 (lane i receives)");
    hipLaunchKernelGGL((map_kernel<0xC000 | (1 << 10) | (8 << 5)>), 1, 64, 0, 0, d); show("ROTATE dir 1, n 8 (lane i receives)");
    double* io; long long* cyc; hipMalloc(&io, 64 * 8 * 4096); hipMalloc(&cyc, 8 * 4096); hipMemset(io, 0, 64 * 8 * 4096);
    for (int waves : {1, 4096 * 1}) {
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL((chain_kernel<0>), waves, 64, 0, 0, io, 2000, cyc);
            else hipLaunchKernelGGL((chain_kernel<1>), waves, 64, 0, 0, io, 2000, cyc);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s, %d wavefront(s): %.1f cycles per dependent 64-bit shift + add (%.2f ms)\n", mode ? "ds_swizzle rotate" : "DPP row_shr:1     ", waves, (double)c / (2000 * 8), ms);
        }
    }
    return 0;
}
