// Probe (round 4): issue cost of the 32-bit integer multiplies the compiler uses for index arithmetic, against full-rate instructions, on gfx950.
// Eight independent chains per lane, 4096 wavefronts (four per SIMD): the kernel time per instruction is the issue cost.
// Build + run: hipcc --offload-arch=gfx950 -O2 int_mul_rate.hip -o /tmp/imr && /tmp/imr
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void rate_kernel(unsigned* io, int reps, unsigned m) {
    unsigned x[8];
    double d[8];
    for (int s = 0; s < 8; ++s) { x[s] = io[threadIdx.x] + s; d[s] = (double)x[s]; }
    unsigned long long w = ((unsigned long long)m << 20) + 7;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[s]) : "v"(m));
            if (MODE == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[s]) : "v"(m));
            if (MODE == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[s]) : "v"(m));
            if (MODE == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[s]) : "v"(m));
            if (MODE == 4) { unsigned long long t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=&v"(t) : "v"(x[s]), "v"(m), "v"(w) : "vcc"); x[s] = (unsigned)t; }
            if (MODE == 5) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[s]) : "v"((double)1.0000001));
            if (MODE == 6) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[s]) : "v"(m));
            if (MODE == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[s]) : "v"(m));
        }
    }
    unsigned acc = 0;
    for (int s = 0; s < 8; ++s) acc += x[s] + (unsigned)d[s];
    io[threadIdx.x + blockIdx.x * 64] = acc;
}
int main() {
    unsigned* io; hipMalloc(&io, 64 * 4 * 4096); hipMemset(io, 1, 64 * 4 * 4096);
    const char* names[8] = {"v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_u64_u32", "v_fma_f64", "v_lshl_add_u32", "v_add_u32"};
    const int reps = 20000, waves = 4096;
    float base = 0;
    for (int mode = 7; mode >= 0; --mode) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            switch (mode) {
                case 0: hipLaunchKernelGGL((rate_kernel<0>), waves, 64, 0, 0, io, reps, 40u); break;
                case 1: hipLaunchKernelGGL((rate_kernel<1>), waves, 64, 0, 0, io, reps, 40u); break;
                case 2: hipLaunchKernelGGL((rate_kernel<2>), waves, 64, 0, 0, io, reps, 40u); break;
                case 3: hipLaunchKernelGGL((rate_kernel<3>), waves, 64, 0, 0, io, reps, 40u); break;
                case 4: hipLaunchKernelGGL((rate_kernel<4>), waves, 64, 0, 0, io, reps, 40u); break;
                case 5: hipLaunchKernelGGL((rate_kernel<5>), waves, 64, 0, 0, io, reps, 40u); break;
                case 6: hipLaunchKernelGGL((rate_kernel<6>), waves, 64, 0, 0, io, reps, 40u); break;
                case 7: hipLaunchKernelGGL((rate_kernel<7>), waves, 64, 0, 0, io, reps, 40u); break;
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        if (mode == 7) base = best;
        // 4096 wavefronts over 1024 SIMDs = 4 per SIMD; instructions per SIMD = 4 * reps * 8
        printf("%-16s %8.3f ms  = %.2f x v_add_u32;  %.2f ns per wavefront instruction per SIMD\n", names[mode], best, best / base,
               best * 1e6 / (4.0 * reps * 8));
    }
    return 0;
}
