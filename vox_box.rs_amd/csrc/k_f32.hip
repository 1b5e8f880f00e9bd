// k_f32.hip -- the Sample = f32 instantiation of the slice traits, REFERENCE-FAITHFUL.
//
// The reference's traits are generic over the Sample type; monomorphised at f32 every operation of
//   Autocorrelate::autocorrelate_mut   src/periodic.rs:276-289   r[lag] = fold(x[0], |acc, (a, b)| acc + a * b)
//   Normalize::normalize               src/waves.rs:60-76
//   LPC::lpc_mut (Levinson)            src/spectrum.rs:63-84
//   LPC::lpc_praat_mut (Burg)          src/spectrum.rs:101-146
//   Pitched::pitch up to the lag curve src/periodic.rs:400-408   autocorrelate -> normalize -> / lag window, all in f32
// rounds to f32, in the order the source states.  A sequential f32 fold cannot be re-associated without changing its
// rounding, so these kernels do NOT parallelise inside a fold: the parallelism is across lags (one lane per lag of one
// frame) or across frames (one lane per frame), every fold runs in the reference's order with FP contraction off, and the
// results are BIT-IDENTICAL to a plain-C f32 restatement of the same statements (tests/test_gpu_f32.py asserts equality).  The wider-and-faster
// forms (f64 arithmetic on the widened frame, rounded once) stay available as vbx_*_f32_wide.
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"
#include "vbx_pitch_refine.hpp"

namespace vbx {

// One wavefront per frame; the (windowed, f32) frame in LDS; lane = lag.  r[lag] = x[0] + sum_{i=1}^{n-lag-1} x[i] x[i+lag]
// as a left fold in f32: x[i] is one broadcast LDS read, x[i + lag] a conflict-free one (consecutive lanes, consecutive words).
__device__ __forceinline__ float autocorr_fold_f32(const float *xs, int n, int lag) {
#pragma clang fp contract(off)
    float acc = xs[0];                                       // the fold's seed (Q1)
    const int m = n - lag;                                   // i runs over [1, m)
    int i = 1;
    for (; i + 4 <= m; i += 4) {
        acc = acc + xs[i] * xs[i + lag];
        acc = acc + xs[i + 1] * xs[i + 1 + lag];
        acc = acc + xs[i + 2] * xs[i + 2 + lag];
        acc = acc + xs[i + 3] * xs[i + 3 + lag];
    }
    for (; i < m; i++) acc = acc + xs[i] * xs[i + lag];
    return acc;
}

__device__ __forceinline__ void load_frame_f32(float *xs, const float *__restrict__ xf, const float *__restrict__ window, int n, int lane) {
#pragma clang fp contract(off)
    for (int i = lane; i < n; i += 64) xs[i] = (window != nullptr) ? xf[i] * window[i] : xf[i];   // Windower<f32>: an f32 product
}

__global__ __launch_bounds__(64) void autocorr_f32_exact_kernel(const float *__restrict__ x, long n_frames, int n, long stride,
                                                                const float *__restrict__ window, int n_lags, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float xs32[];
    const long f = xcd_item(blockIdx.x, n_frames);
    if (f >= n_frames) return;
    const int lane = lane_id();
    load_frame_f32(xs32, x + f * stride, window, n, lane);
    wave_sync();
    for (int lag0 = 0; lag0 < n_lags; lag0 += 64) {
        const int lag = lag0 + lane;
        if (lag < n_lags) out[f * (long)n_lags + lag] = autocorr_fold_f32(xs32, n, lag < n ? lag : n - 1);
    }
}

// Pitched<f32, f32>::pitch: the lag curve in f32 exactly as the reference computes it (autocorrelate(n) -> normalize -> each
// entry / the lag window's entry, :403-408), widened, then the shared refinement with T = f32 roundings (pitch_params_t::f32).
__global__ __launch_bounds__(64) void pitch_f32_exact_kernel(const float *__restrict__ x, long n_frames, int n, long stride,
                                                             const float *__restrict__ window, const float *__restrict__ lag_window32,
                                                             pitch_params_t pp, int ys_off /* bytes */, double *__restrict__ out_cand, long cand_ld,
                                                             int32_t *__restrict__ out_count, int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) float xs32[];
    const long f = xcd_item(blockIdx.x, n_frames);
    if (f >= n_frames) return;
    const int lane = lane_id();
    float *lg = xs32 + ((n + 3) & ~3);                       // the lag curve, f32
    double *ys = reinterpret_cast<double *>(reinterpret_cast<char *>(xs32) + ys_off);
    load_frame_f32(xs32, x + f * stride, window, n, lane);
    wave_sync();
    float m = -1.0f;                                         // max_amplitude over all n lags (src/waves.rs:44-58; NaN never wins)
    for (int lag0 = 0; lag0 < n; lag0 += 64) {
        const int lag = lag0 + lane;
        if (lag < n) {
            const float r = autocorr_fold_f32(xs32, n, lag);
            lg[lag] = r;
            const float a = (r < 0.0f) ? r * -1.0f : r;      // Amplitude::amplitude on f32
            m = (a > m) ? a : m;
        }
    }
    m = (float)wave_max((double)m);
    wave_sync();
    { const float first = lg[0]; if (first != first) m = first; }        // the fold starts from |r[0]|: a NaN there stays (:47-48)
    {
#pragma clang fp contract(off)
        const float scale = 1.0f / m;                        // :69-71
        for (int i = lane; i < n; i += 64) ys[i] = (double)((lg[i] * scale) / lag_window32[i]);   // :404-408, then widened
    }
    if (lane < Y_PAD) ys[n + lane] = 0.0;
    wave_sync();
    double2 *full = pp.full_off > 0 ? reinterpret_cast<double2 *>(reinterpret_cast<char *>(xs32) + pp.full_off) : nullptr;
    pitch_refine_store(ys, n, pp, f, out_cand, cand_ld, out_count, status, nullptr, 0.0, full);
}

// LPC::lpc_mut at T = f32 (src/spectrum.rs:63-84): one lane per row, every operation an f32 operation in source order.
__global__ void levinson_f32_exact_kernel(const float *__restrict__ r, long n_rows, long r_stride, int p,
                                          float *__restrict__ out, long out_ld, float *__restrict__ out_kc) {
#pragma clang fp contract(off)
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const float *rr = r + row * r_stride;
    float ac[VBX_MAX_LPC_ORDER_K + 1], tmp[VBX_MAX_LPC_ORDER_K + 1];
    float err = rr[0];
    ac[0] = 1.0f;
    for (int i = 1; i <= p; i++) ac[i] = 0.0f;
    for (int i = 1; i <= p; i++) {
        float acc = rr[i];
        for (int j = 1; j < i; j++) acc = acc + (ac[j] * rr[i - j]);
        const float k = -acc / err;
        ac[i] = k;
        if (out_kc != nullptr) out_kc[row * (long)p + (i - 1)] = k;
        for (int j = 0; j < p; j++) tmp[j] = ac[j];
        for (int j = 1; j < i; j++) ac[j] = ac[j] + (k * tmp[i - j]);
        err = err * (1.0f - (k * k));
    }
    for (int i = 0; i <= p; i++) out[row * out_ld + i] = ac[i];
}

// LPC::lpc_praat_mut at T = f32 (src/spectrum.rs:101-146): the two sums of every order are sequential f32 folds over the
// frame, so the parallelism is one LANE per frame; b1 / b2 live in a global scratch laid out [block][2][j][lane] (a wave's
// accesses are coalesced).  Rates are those of a reference path, not of the f64 kernel (k_burg.hip).
__global__ __launch_bounds__(64) void burg_f32_exact_kernel(const float *__restrict__ x, long f0, long n_frames, int n, long stride,
                                                            const float *__restrict__ window, int p, float *__restrict__ out,
                                                            int32_t *__restrict__ status, float *__restrict__ scratch) {
#pragma clang fp contract(off)
    const int lane = lane_id();
    const long f = f0 + (long)blockIdx.x * 64 + lane;
    const bool have = f < n_frames;
    const float *xf = x + (have ? f : n_frames - 1) * stride;
    float *b1 = scratch + ((long)blockIdx.x * 2 + 0) * (long)n * 64 + lane;      // element j at b1[j * 64]
    float *b2 = scratch + ((long)blockIdx.x * 2 + 1) * (long)n * 64 + lane;
    auto sample = [&](int j) -> float { return (window != nullptr) ? xf[j] * window[j] : xf[j]; };
    // :108-114  b1[0] = x[0]; b2[n-2] = x[n-1]; b1[j-1] = b2[j-2] = x[j-1] for j in 2..n
    for (int j = 0; j < n; j++) { b1[(long)j * 64] = 0.0f; b2[(long)j * 64] = 0.0f; }
    b1[0] = sample(0);
    b2[(long)(n - 2) * 64] = sample(n - 1);
    for (int j = 2; j < n; j++) { const float v = sample(j - 1); b1[(long)(j - 1) * 64] = v; b2[(long)(j - 2) * 64] = v; }
    float aa[VBX_MAX_LPC_ORDER_K], co[VBX_MAX_LPC_ORDER_K];
    for (int t = 0; t < p; t++) { aa[t] = 0.0f; co[t] = 0.0f; }
    int st = 0;
    for (int i = 1; i <= p && st == 0; i++) {
        float num = 0.0f, denum = 0.0f;
        for (int j = 1; j + i < n + 1; j++) {
            const float u = b1[(long)(j - 1) * 64], v = b2[(long)(j - 1) * 64];
            num = num + u * v;
            denum = denum + u * u + v * v;                   // (denum + b1^2) + b2^2, :120
        }
        if (denum <= 0.0f) { st = 1; break; }                // Err(LPC), :123-125
        co[i - 1] = 2.0f * num / denum;
        for (int j = 1; j < i; j++) co[j - 1] = aa[j - 1] - co[i - 1] * aa[i - j - 1];
        if (i < p) {
            for (int j = 1; j < i + 1; j++) aa[j - 1] = co[j - 1];
            const float a = aa[i - 1];
            for (int j = 1; j + i < n; j++) {
                const float u = b1[(long)(j - 1) * 64], v = b2[(long)(j - 1) * 64];
                const float un = b1[(long)j * 64], vn = b2[(long)j * 64];
                b1[(long)(j - 1) * 64] = u - a * v;
                b2[(long)(j - 1) * 64] = vn - a * un;        // uses the OLD b1[j] (it is overwritten in the next iteration), :137
            }
        }
    }
    if (have) {
        for (int t = 0; t < p; t++) out[f * (long)p + t] = (st == 0) ? co[t] * -1.0f : 0.0f;   // :142-144
        if (status != nullptr) status[f] = st;
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------

void launch_autocorr_f32_exact(hipStream_t s, const float *x, long F, int n, long stride, const float *window, int n_lags, float *out) {
    hipLaunchKernelGGL(autocorr_f32_exact_kernel, dim3((unsigned)F), dim3(64), ((size_t)n + 64) * sizeof(float), s,
                       x, F, n, stride, window, n_lags, out);
}

size_t pitch_f32_exact_lds_bytes(int n, int kmax) {
    const size_t head = (size_t)2 * ((n + 3) & ~3) * sizeof(float);
    return ((head + 15) & ~(size_t)15) + ((pitch_refine_lds_bytes(n) + 15) & ~15) + pitch_full_list_bytes(n, kmax);
}

void launch_pitch_f32_exact(hipStream_t s, const float *x, long F, int n, long stride, const float *window, const float *lag_window32,
                            double sample_rate, double threshold, double fmin, double fmax, int kmax, pitch_t *out_cand, long cand_ld,
                            int32_t *out_count, int32_t *status) {
    const size_t head = (((size_t)2 * ((n + 3) & ~3) * sizeof(float)) + 15) & ~(size_t)15;
    const size_t refine = (pitch_refine_lds_bytes(n) + 15) & ~15, extra = pitch_full_list_bytes(n, kmax);
    pitch_params_t pp;
    pp.sample_rate = sample_rate; pp.threshold = threshold; pp.fmin = fmin; pp.fmax = fmax; pp.kmax = kmax;
    pp.full_off = extra ? (int)(head + refine) : 0;
    pp.f32 = 1;
    hipLaunchKernelGGL(pitch_f32_exact_kernel, dim3((unsigned)F), dim3(64), head + refine + extra, s,
                       x, F, n, stride, window, lag_window32, pp, (int)head, reinterpret_cast<double *>(out_cand), cand_ld, out_count, status);
}

void launch_levinson_f32_exact(hipStream_t s, const float *r, long rows, long r_stride, int p, float *out, long out_ld, float *out_kc) {
    hipLaunchKernelGGL(levinson_f32_exact_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, r, rows, r_stride, p, out, out_ld, out_kc);
}

size_t burg_f32_exact_scratch_bytes(long frames, int n) { return (size_t)((frames + 63) / 64) * 2 * (size_t)n * 64 * sizeof(float); }

void launch_burg_f32_exact(hipStream_t s, const float *x, long f0, long f1, long F, int n, long stride, const float *window, int p,
                           float *out, int32_t *status, float *scratch) {
    const long cnt = f1 - f0;
    hipLaunchKernelGGL(burg_f32_exact_kernel, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, s,
                       x, f0, (f1 < F ? f1 : F), n, stride, window, p, out, status, scratch);
}

}  // namespace vbx
