// mpc_order.hpp -- dispatch order of the problems of a batch (MPCGPU_OPT_ORDER).
//
// One wavefront solves one problem from start to end and the hardware hands out workgroups in index order.  Solves differ in
// length (10^1 .. 10^4 PANOC steps); whatever is long and starts LAST finishes on a draining GPU -- at 8192 problems per launch
// a quarter of the kernel time (profiles/archive/r03_order_ab.txt).  In a receding-horizon loop problem i of this call is robot i one
// tick later: how long it took LAST time is a good guess of how long it takes now.  The library keeps the psi-evaluation counts
// of its previous call anyway (mpcgpu_last_eval_counts); three small kernels turn them into a permutation, longest first
// (counting sort on 1024 coarse bins: the order inside a bin is whatever the atomics give, which changes nothing -- every
// problem is solved independently and writes the outputs of ITS index, so results are bitwise those of the order as given).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcgpu {

constexpr int ORD_BINS = 1024;

__device__ __forceinline__ int order_bin(int32_t n_eval, int shift) {
    const int b = n_eval >> shift;
    return b < 0 ? 0 : (b >= ORD_BINS ? ORD_BINS - 1 : b);
}
// evals: [B][2] (psi evaluations, of those with gradient) of the previous call
__global__ __launch_bounds__(256) void order_hist_kernel(const int32_t* __restrict__ evals, int B, int* __restrict__ bins, int shift) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) atomicAdd(&bins[order_bin(evals[2 * b], shift)], 1);
}
// counts -> first slot of every bin when the bins are laid out from the LAST (longest) to the first
__global__ __launch_bounds__(ORD_BINS) void order_scan_kernel(int* __restrict__ bins) {
    __shared__ int s[ORD_BINS];
    const int t = threadIdx.x;
    const int mine = bins[ORD_BINS - 1 - t];   // t = 0: the last bin
    s[t] = mine;
    __syncthreads();
    for (int d = 1; d < ORD_BINS; d <<= 1) {
        const int v = t >= d ? s[t - d] : 0;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    bins[ORD_BINS - 1 - t] = s[t] - mine;      // exclusive: problems in longer bins
}
__global__ __launch_bounds__(256) void order_scatter_kernel(const int32_t* __restrict__ evals, int B, int* __restrict__ bins,
                                                            int32_t* __restrict__ perm, int shift) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) perm[atomicAdd(&bins[order_bin(evals[2 * b], shift)], 1)] = b;
}

}  // namespace mpcgpu
