// k_burg.hip -- Burg LPC (LPC::lpc_praat_mut, src/spectrum.rs:101-146, Q12).
//
// One wavefront per frame.  Lane l keeps b1/b2 elements [l*EPL, (l+1)*EPL) in registers, so
// the reference's shift  b2[j] <- b2[j+1] - a*b1[j+1]  needs ONE neighbour-lane fetch per order
// (DPP wave_shl) instead of a pass through memory.  Per order: two wave reductions (num, denum),
// the coefficient recursion (tiny, LDS-resident, lane 0), one fused update sweep.
// The shrinking valid range [0, N-i) is kept by zeroing exactly the element that drops out.
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

template <int EPL>
__global__ __launch_bounds__(64) void burg_kernel(
    const double *__restrict__ x, long n_frames, int n, long stride, const double *__restrict__ window,
    int p, double *__restrict__ out, int32_t *__restrict__ status) {
    __shared__ double aa[VBX_MAX_LPC_ORDER_K + 2];
    __shared__ double co[VBX_MAX_LPC_ORDER_K + 2];
    const long f = blockIdx.x;
    if (f >= n_frames) return;
    const int lane = lane_id();
    const double *xf = x + f * stride;

    double b1[EPL], b2[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        const int j = lane * EPL + e;
        double v = (j < n) ? xf[j] : 0.0;
        if (window != nullptr && j < n) v *= window[j];
        b1[e] = v;
    }
    // b2[j] = x[j+1]  (zero past the frame);  b1[j] = x[j] for j <= n-2  (src/spectrum.rs:108-114)
    {
        const double nxt = from_next_lane(b1[0]);
#pragma unroll
        for (int e = 0; e < EPL - 1; e++) b2[e] = b1[e + 1];
        b2[EPL - 1] = nxt;
        const int last = n - 1;
        const int kb = last % EPL, lb = last / EPL;
#pragma unroll
        for (int e = 0; e < EPL; e++) if (e == kb && lane == lb) b1[e] = 0.0;
    }

    int st = 0;
    for (int i = 1; i <= p; i++) {
        double num = 0.0, den = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            num = fma(b1[e], b2[e], num);
            den = fma(b1[e], b1[e], den);
            den = fma(b2[e], b2[e], den);
        }
        num = wave_sum(num);
        den = wave_sum(den);
        if (den <= 0.0) { st = 1; break; }          // Err(LPC), src/spectrum.rs:123-125 (NaN falls through)
        const double c = 2.0 * num / den;
        __syncthreads();
        if (lane == 0) {
            co[i - 1] = c;
            for (int j = 1; j < i; j++) co[j - 1] = aa[j - 1] - c * aa[i - j - 1];
        }
        __syncthreads();
        if (i < p) {
            if (lane == 0) for (int j = 1; j <= i; j++) aa[j - 1] = co[j - 1];
            const double a = c;                      // aa[i-1] == coeffs[i-1]
            const double nb1 = from_next_lane(b1[0]);
            const double nb2 = from_next_lane(b2[0]);
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const double b1n = (e + 1 < EPL) ? b1[e + 1] : nb1;   // old b1[j+1]
                const double b2n = (e + 1 < EPL) ? b2[e + 1] : nb2;   // old b2[j+1]
                const double t1 = fma(-a, b2[e], b1[e]);
                const double t2 = fma(-a, b1n, b2n);
                b1[e] = t1;
                b2[e] = t2;
            }
            // element n-i-1 leaves the valid range (the update loop runs j-1 < n-i-1)
            const int drop = n - i - 1;
            if (drop >= 0) {
                const int kb = drop % EPL, lb = drop / EPL;
#pragma unroll
                for (int e = 0; e < EPL; e++) if (e == kb && lane == lb) { b1[e] = 0.0; b2[e] = 0.0; }
            }
        }
    }
    __syncthreads();
    if (lane < p) out[f * (long)p + lane] = (st == 0) ? co[lane] * -1.0 : 0.0;   // src/spectrum.rs:142-144
    if (status != nullptr && lane == 0) status[f] = st;
}

bool burg_supported(int n, int p) {
    return n >= 2 && n <= 64 * 64 && p >= 1 && p <= VBX_MAX_LPC_ORDER_K;
}

void launch_burg(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                 int p, double *out, int32_t *status) {
    dim3 g((unsigned)F), b(64);
#define VBX_BURG(E) hipLaunchKernelGGL((burg_kernel<E>), g, b, 0, s, x, F, n, stride, window, p, out, status)
    if (n <= 64 * 8) VBX_BURG(8);
    else if (n <= 64 * 16) VBX_BURG(16);
    else if (n <= 64 * 20) VBX_BURG(20);
    else if (n <= 64 * 32) VBX_BURG(32);
    else VBX_BURG(64);
#undef VBX_BURG
}

}  // namespace vbx
