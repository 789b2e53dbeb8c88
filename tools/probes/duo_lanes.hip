// Probe: the cross-lane primitives of the two-problems-per-wavefront layout (rows 0-1 = problem A, rows 2-3 = problem B) on
// gfx950, with the whole wave active and with only ONE half active (divergent state machine):
//   (1) xor butterfly inside a 16-lane row (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror):
//       every lane of a row must end with bitwise the same total;
//   (2) v_permlane16_swap as the row exchange inside each half -> per-half total in all 32 lanes;
//   (3) row_newbcast:0 (first lane of the row to the whole row) for the suffix-scan cascade;
//   (4) the same under a partial EXEC mask (upper half only / lower half only).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/duo_lanes tools/probes/duo_lanes.hip && /tmp/duo_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp0(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_allsum(double x) {
    x += dpp0<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp0<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp0<0x141>(x);  // row_half_mirror
    x += dpp0<0x140>(x);  // row_mirror
    return x;
}
// rows (0,1) and (2,3) exchanged: returns x of the OTHER row of the same half, lane for lane
__device__ __forceinline__ void row_pair(double x, double& even_rows, double& odd_rows) {
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    even_rows = __hiloint2double(b[0], a[0]);  // [r0, r0, r2, r2]
    odd_rows = __hiloint2double(b[1], a[1]);   // [r1, r1, r3, r3]
}
__device__ __forceinline__ double half_allsum(double x) {
    x = row_allsum(x);
    double e, o;
    row_pair(x, e, o);
    return e + o;
}
// inclusive suffix sum over the 32 lanes of each half
__device__ __forceinline__ double half_suffix(double x) {
    x += dpp0<0x101>(x); x += dpp0<0x102>(x); x += dpp0<0x104>(x); x += dpp0<0x108>(x);  // row_shl 1,2,4,8
    double e, o;
    row_pair(x, e, o);
    return x + dpp0<0x150, 0x5>(o);  // rows 0/2 += first lane of rows 1/3 (row_newbcast:0), rows 1/3 += 0
}

__global__ void probe(const double* in, double* out, int mode) {
    const int lane = threadIdx.x;
    const double x = in[lane];
    double s = -1.0, sf = -1.0;
    const bool on = mode == 0 || (mode == 1 && lane >= 32) || (mode == 2 && lane < 32);
    if (on) {
        s = half_allsum(x);
        sf = half_suffix(x);
    }
    out[lane] = s;
    out[64 + lane] = sf;
}

int main() {
    double h_in[64], h_out[128];
    for (int i = 0; i < 64; ++i) h_in[i] = std::sin(1.0 + 0.37 * i) * std::pow(10.0, (i % 7) - 3);
    double *d_in, *d_out;
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    int bad = 0;
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, mode);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        for (int half = 0; half < 2; ++half) {
            const bool on = mode == 0 || (mode == 1 && half == 1) || (mode == 2 && half == 0);
            double ref = 0.0;
            for (int i = 0; i < 32; ++i) ref += h_in[32 * half + i];
            bool same = true, sfx = true;
            for (int i = 0; i < 32; ++i) {
                same &= h_out[32 * half + i] == h_out[32 * half];
                double r = 0.0;
                for (int j = 31; j >= i; --j) r += h_in[32 * half + j];
                sfx &= std::fabs(h_out[64 + 32 * half + i] - r) <= 1e-12 * (1.0 + std::fabs(r));
            }
            const bool okv = on ? (same && std::fabs(h_out[32 * half] - ref) <= 1e-12 * (1.0 + std::fabs(ref)) && sfx)
                                : (h_out[32 * half] == -1.0 && h_out[64 + 32 * half] == -1.0);
            printf("mode %d half %d (%s): total %.17g (host %.17g) identical in all lanes: %d, suffix ok: %d -> %s\n", mode, half,
                   on ? "active" : "masked", h_out[32 * half], ref, (int)same, (int)sfx, okv ? "OK" : "FAIL");
            bad += !okv;
        }
    }
    printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad;
}
