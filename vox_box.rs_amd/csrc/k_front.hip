// k_front.hip -- the steps either side of the hot path:
//   PCM ingestion   i16 -> f64 / 32767  (hound reader as used at tests/lib.rs:17-19)
//   RMS::rms        src/waves.rs:10-23
//   Filter::preemphasis  src/waves.rs:82-96: backwards recurrence y[i] = x[i] + c*y[i+1], c = 2*pi*factor
// All three are HBM-bound byte/stream work: coalesced loads, one pass.
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

// two samples per lane and step: one 4-B load, one 16-B store -- every wave instruction touches one
// contiguous span (256 B in, 1 KiB out)
__global__ void pcm16_kernel(const int16_t *__restrict__ pcm, size_t n, double denom, double *__restrict__ out) {
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const bool aligned = ((reinterpret_cast<uintptr_t>(pcm) & 3) | (reinterpret_cast<uintptr_t>(out) & 15)) == 0;
    const size_t n2 = aligned ? n / 2 : 0;
    const double r = 1.0 / denom;
    for (size_t v = t0; v < n2; v += step) {
        const int w = reinterpret_cast<const int *>(pcm)[v];
        const double lo = (double)(short)(w & 0xffff), hi = (double)(short)(w >> 16);
        reinterpret_cast<double2 *>(out)[v] = make_double2(div_exact_small(lo, denom, r), div_exact_small(hi, denom, r));
    }
    for (size_t i = n2 * 2 + t0; i < n; i += step) out[i] = div_exact_small((double)pcm[i], denom, r);
}

// rms: one wavefront per frame
__global__ __launch_bounds__(64) void rms_kernel(const double *__restrict__ x, long n_frames, int n, long stride,
                                                 const double *__restrict__ window, double *__restrict__ out) {
    const long f = xcd_item(blockIdx.x, n_frames);
    if (f >= n_frames) return;
    const int lane = lane_id();
    const double *xf = x + f * stride;
    double s = 0.0;
    for (int i = lane; i < n; i += 64) {
        double v = xf[i];
        if (window != nullptr) v *= window[i];
        s = fma(v, v, s);
    }
    s = wave_sum(s);
    if (lane == 0) out[f] = sqrt(s / (double)n);
}

// preemphasis: one wavefront per frame; lane l owns elements [l*E, (l+1)*E) (E = ceil(n/64)), kept in LDS.
//   local pass   y_loc[e] = x[e] + c*y_loc[e+1]   (carry-in 0)
//   lane scan    Y_l = y_loc_first(l) + c^E * Y_{l+1}    (backward Hillis-Steele over the 64 lanes)
//   fix-up       y[e] = y_loc[e] + c^(E-e) * Y_{l+1}
__global__ __launch_bounds__(64) void preemphasis_kernel(const double *__restrict__ x, long n_frames, int n, long stride,
                                                         double c, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long f = xcd_item(blockIdx.x, n_frames);
    if (f >= n_frames) return;
    const int lane = lane_id();
    const double *xf = x + f * stride;
    double *yo = out + f * (long)n;
    const int E = (n + 63) / 64;
    for (int i = lane; i < 64 * E; i += 64) smem[i] = (i < n) ? xf[i] : 0.0;
    __syncthreads();
    if (!(fabs(c) < 1.0)) {
        // |2 pi factor| >= 1: the recurrence grows like c^n, and the powers c^(E 2^k) of the lane scan overflow long
        // before the reference's sequential values do (inf * 0 would poison finite entries with NaN).  Unstable
        // filters take the reference's own order: one lane walks the frame backwards (src/waves.rs:88-94).
        if (lane == 0) {
            double carry = 0.0;
            for (int i = n - 1; i >= 0; i--) { carry = fma(c, carry, smem[i]); smem[i] = carry; }
        }
        __syncthreads();
        for (int i = lane; i < n; i += 64) yo[i] = smem[i];
        return;
    }
    double *mine = smem + lane * E;
    double carry = 0.0;
    for (int e = E - 1; e >= 0; e--) { carry = fma(c, carry, mine[e]); mine[e] = carry; }
    // A = c^E
    double A = 1.0;
    for (int e = 0; e < E; e++) A *= c;
    double Y = carry;                                   // value at the lane's first element, carry-in 0
    double Ad = A;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double other = __shfl_down(Y, d, 64);
        if (lane + d < 64) Y = fma(Ad, other, Y);
        Ad *= Ad;
    }
    const double cin = __shfl_down(Y, 1, 64);           // Y_{l+1}
    const double carry_in = (lane < 63) ? cin : 0.0;
    double pw = c;                                      // c^(E-e) for e = E-1 .. 0
    for (int e = E - 1; e >= 0; e--) { mine[e] = fma(pw, carry_in, mine[e]); pw *= c; }
    __syncthreads();
    for (int i = lane; i < n; i += 64) yo[i] = smem[i];
}

// resample front end of find_formants (src/lib.rs:57-61): sample 0.10's Linear + Converter reduce to
//   out[k] = (x[li+1] - x[li]) * frac + x[li]   with (li, frac) a frame-independent table built on the host by
// the crate's own recurrence (interpolation_value += 1/ratio; whole steps advance the source), zeros past the end.
__global__ void resample_kernel(const double *__restrict__ x, long n_frames, int n, long stride,
                                const int32_t *__restrict__ tab_idx, const double *__restrict__ tab_frac, int m,
                                double *__restrict__ out) {
#pragma clang fp contract(off)   // (diff * value) + left, two roundings as on the CPU
    const long total = n_frames * (long)m;
    const long step = (long)gridDim.x * blockDim.x;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += step) {
        const long f = idx / m;
        const int k = (int)(idx - f * m);
        const int li = tab_idx[k];
        const double *xf = x + f * stride;
        const double left = (li < n) ? xf[li] : 0.0;
        const double right = (li + 1 < n) ? xf[li + 1] : 0.0;
        const double diff = right - left;
        out[idx] = (diff * tab_frac[k]) + left;
    }
}

// VecDeque view (src/periodic.rs:291-304): logical sample i of the deque is ring[(head + i) % capacity].
// Frame t of the Windower view over the deque, copied into a dense [F, N] batch (coalesced on the write side).
__global__ void ring_frames_kernel(const double *__restrict__ ring, long capacity, long head, long n_frames, int n,
                                   long stride, double *__restrict__ out) {
    const long total = n_frames * (long)n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long t = e / n, i = e - t * n;
        out[e] = ring[(head + t * stride + i) % capacity];
    }
}

void launch_ring_frames(hipStream_t s, const double *ring, long capacity, long head, long F, int n, long stride, double *out) {
    long blocks = (F * (long)n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(ring_frames_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ring, capacity, head, F, n, stride, out);
}

void launch_resample(hipStream_t s, const double *x, long F, int n, long stride, const int32_t *tab_idx,
                     const double *tab_frac, int m, double *out) {
    long blocks = (F * (long)m + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, F, n, stride, tab_idx, tab_frac, m, out);
}

void launch_pcm16(hipStream_t s, const int16_t *pcm, size_t n, double denom, double *out) {
    size_t blocks = (n / 2 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pcm, n, denom, out);
}

void launch_rms(hipStream_t s, const double *x, long F, int n, long stride, const double *window, double *out) {
    hipLaunchKernelGGL(rms_kernel, dim3((unsigned)F), dim3(64), 0, s, x, F, n, stride, window, out);
}

void launch_preemphasis(hipStream_t s, const double *x, long F, int n, long stride, double c, double *out) {
    const size_t lds = (size_t)(64 * ((n + 63) / 64)) * sizeof(double);
    hipLaunchKernelGGL(preemphasis_kernel, dim3((unsigned)F), dim3(64), lds, s, x, F, n, stride, c, out);
}

// f32 instantiation helpers: the frames of a (strided, optionally windowed) f32 batch as a dense f64 batch -- the windowed
// product rounded to f32 first, as the reference's Windower<f32> produces it -- and f64 result rows rounded to f32.
__global__ void widen_frames_kernel(const float *__restrict__ x, long n_frames, int n, long stride,
                                    const float *__restrict__ window, double *__restrict__ out) {
    const long total = n_frames * (long)n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long t = e / n; const int i = (int)(e - t * n);
        float v = x[t * stride + i];
        if (window != nullptr) v = v * window[i];
        out[e] = (double)v;
    }
}

__global__ void narrow_kernel(const double *__restrict__ in, long count, float *__restrict__ out) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) out[e] = (float)in[e];
}

void launch_widen_frames(hipStream_t s, const float *x, long F, int n, long stride, const float *window, double *out) {
    long blocks = (F * (long)n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(widen_frames_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, F, n, stride, window, out);
}

void launch_narrow(hipStream_t s, const double *in, long count, float *out) {
    long blocks = (count + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(narrow_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, count, out);
}

}  // namespace vbx
