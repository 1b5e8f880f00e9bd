// k_selftest.hip -- on-device check of the wave64 cross-lane helpers (DPP forms against the
// ds_bpermute forms whose semantics HIP defines).  Read back by tests/test_gpu_selftest.py.
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

// out[lane*8 + j]: 0 from_next_lane, 1 __shfl_down ref, 2 from_prev_lane, 3 __shfl_up ref,
//                  4 wave_sum (DPP), 5 wave_sum (shfl), 6 wave_max, 7 readlane(17)
// out[512 + lane*8 + j]: group_sum<4,8,16,32,64> in j = 0..4, wave_inclusive_scan in j = 5
__global__ __launch_bounds__(64) void selftest_kernel(double *__restrict__ out) {
    const int lane = lane_id();
    const double v = 1.0 + 0.37 * (double)lane + 1e-3 * (double)((lane * 7919) % 64);
    double *o = out + lane * 8;
    o[0] = from_next_lane(v);
    const double dn = __shfl_down(v, 1, 64);
    o[1] = (lane == 63) ? 0.0 : dn;
    o[2] = from_prev_lane(v);
    const double up = __shfl_up(v, 1, 64);
    o[3] = (lane == 0) ? 0.0 : up;
    o[4] = wave_sum(v);
    o[5] = wave_sum_shfl(v);
    o[6] = wave_max(v);
    o[7] = readlane_f64(v, 17);
    double *g = out + 512 + lane * 8;
    g[0] = group_sum<4>(v); g[1] = group_sum<8>(v); g[2] = group_sum<16>(v);
    g[3] = group_sum<32>(v); g[4] = group_sum<64>(v); g[5] = wave_inclusive_scan(v); g[6] = 0.0; g[7] = 0.0;
}

void launch_selftest(hipStream_t s, double *out) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, s, out);
}

}  // namespace vbx
