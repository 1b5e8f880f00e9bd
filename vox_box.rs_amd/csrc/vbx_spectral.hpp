// vbx_spectral.hpp -- what the FFT-based analysis kernels share (k_spectral.hip: complex length 1200 = 20 * 20 * 3;
// vbx_spectral_pow2.hpp: 1024 = 16 * 16 * 4, 2048 = 16 * 16 * 8 and 4096 = 16 * 16 * 16): the kernel arguments, what an
// instantiation computes (MODE), the radix-4 butterfly, the exact tail of a padded frame's lag curve and the register
// Levinson recursion.
#pragma once

#include "vbx_device.hpp"
#include "vbx_kernels.hpp"
#include "vbx_pitch_refine.hpp"

namespace vbx {

// what a spectral kernel instantiation computes.  SP_MFCC_HALF (power-of-two kernels): MFCC::mfcc of a frame of 2 Nc samples --
// the frame itself is the real sequence of the transform (no padding), its bins are the transform's bins
enum { SP_ANALYZE = 0, SP_MFCC_ONLY = 1, SP_AC_ONLY = 2, SP_MFCC_HALF = 3, SP_ANALYZE_INTERP = 4, SP_ANALYZE_SPLIT = 5, SP_ANALYZE_INTERP_SPLIT = 6,
       SP_MFCC_ONLY_INTERP = 7 };                          // MFCC::mfcc alone of a padded frame whose bins are interpolated (one transform instead of the chirp-z kernel's two)
// SP_ANALYZE_SPLIT (+ _INTERP_SPLIT): the fused analysis WITHOUT the refinement -- the normalised lag curve goes to a scratch row in
// HBM and scan_curve_kernel / refine_list_kernel / refine_far_kernel (k_spectral_pow2.hip) take it from there.  For the 4096-point
// plan: its 35 KB exchange buffer holds a CU to four frames, i.e. four refining wavefronts, one per SIMD, and a dependent FP64 chain
// alone on a SIMD runs at a third of the rate three of them reach together (measured: 30 ns per frame of refinement against 16
// at 1200 / 2048 samples; split: 21 + 4 + 2).
__host__ __device__ constexpr bool sp_is_interp(int mode) { return mode == SP_ANALYZE_INTERP || mode == SP_ANALYZE_INTERP_SPLIT || mode == SP_MFCC_ONLY_INTERP; }
__host__ __device__ constexpr bool sp_is_mfcc_only(int mode) { return mode == SP_MFCC_ONLY || mode == SP_MFCC_HALF || mode == SP_MFCC_ONLY_INTERP; }
__host__ __device__ constexpr bool sp_is_split(int mode) { return mode == SP_ANALYZE_SPLIT || mode == SP_ANALYZE_INTERP_SPLIT; }
__host__ __device__ constexpr bool sp_is_analyze(int mode) { return mode == SP_ANALYZE || mode == SP_ANALYZE_INTERP || mode == SP_ANALYZE_SPLIT || mode == SP_ANALYZE_INTERP_SPLIT; }
// SP_ANALYZE_INTERP: SP_ANALYZE of a frame whose length does not divide the transform's, with MFCC::mfcc's bins -- samples of
// the frame's DTFT at k / n, between the transform's bins j / M -- interpolated from the transform (mfcc_interp_t, below).

constexpr double SP_UNC_EPS = 6.0 * 400.0 * 2.220446049250313e-16;   // 1 / min w_lag * margin * eps

// a / b, IEEE-rounded, for a divisor that comes from a TABLE with its correctly rounded reciprocal y = RN(1 / b) beside it
// (round 6): q0 = RN(a y), rem = RN(a - q0 b) (one FMA), q = RN(q0 + rem y).  q is the rounding of a value within 2^-104 (relative) of
// a / b, i.e. RN(a / b) itself unless a / b lies that close to a midpoint of two doubles -- no such case in 82,720 trials against exact
// rational arithmetic on the lag windows' own values, and every output of 1.2 M frames at seven shapes bit for bit the IEEE-division
// build's (profiles/r06_headline).  Three vector instructions instead of the ~14 issue slots of v_div_scale / v_rcp / v_div_fmas /
// v_div_fixup: the lag-window divide of src/periodic.rs:406-408 is 22 divisions per lane of the headline kernel.  The callers keep
// the IEEE division for frames whose scale 1 / max|r| is not a normal finite number (a quotient that overflows must stay an infinity).
constexpr int SP_FLAG_PCM = 1, SP_FLAG_LAG_RCP = 2, SP_FLAG_MFCC_DEFER = 4;      // MFCC_DEFER: mfcc_tail_q leaves the filter sums in the row (mfcc_rows_kernel finishes)
__host__ __device__ constexpr int lag_rcp_offset(int n) { return (n + 1) & ~1; }      // the reciprocals follow the window's n entries, 16-byte aligned
__device__ __forceinline__ double quotient_by_table(double a, double b, double y) {
    const double q0 = a * y;
    const double rem = fma(-q0, b, a);
    return fma(rem, y, q0);
}

// ---- small DFTs on separate re / im registers (forward: e^{-i...}) -------------------------------------------------
__device__ __forceinline__ void dft4(double &r0, double &i0, double &r1, double &i1, double &r2, double &i2,
                                     double &r3, double &i3) {
    const double t0r = r0 + r2, t0i = i0 + i2, t1r = r0 - r2, t1i = i0 - i2;
    const double t2r = r1 + r3, t2i = i1 + i3, t3r = r1 - r3, t3i = i1 - i3;
    r0 = t0r + t2r; i0 = t0i + t2i;
    r2 = t0r - t2r; i2 = t0i - t2i;
    r1 = t1r + t3i; i1 = t1i - t3r;          // t1 - i t3
    r3 = t1r - t3i; i3 = t1i + t3r;          // t1 + i t3
}

struct spectral_args_t {
    const double *frames; long n_frames; long stride; const double *window; const double *lag_window;
    const double2 *tab;
    int n;                                                   // frame length (<= the plan's complex FFT length)
    pitch_params_t pp;
    double *out_cand; long cand_ld; int32_t *out_count; int32_t *pitch_status; unsigned long long *work;
    double *out_lpc; long lpc_ld;
    double *out_mfcc; long mfcc_ld; int32_t *mfcc_status;
    const int32_t *bins; const double *slopes; const double *dct; int num_coeffs; int nb;
    int32_t *unsure_list; int32_t *unsure_count;
    double *out_r; int n_lags;                               // SP_AC_ONLY: [F, n_lags] lag sums
    int mfcc_q;                                              // the frame's DFT bin k' is the transform's bin mfcc_q * k' (M / n)
    int pcm;                                                 // bit 0: `frames` points to int16 PCM samples (widened in registers:
                                                             // s / 32767, vbx_device.hpp pcm16_value); full frames only
                                                             // bit 1 (SP_FLAG_LAG_RCP): lag_window[lag_rcp_offset(n) + i] = RN(1 / lag_window[i]) (quotient_by_table)
    mfcc_interp_t ip;                                        // SP_ANALYZE_INTERP
    long f0, n_batch;                                        // power-of-two kernels: this launch covers frames [f0, f0 + n_batch)
    double *curve; long curve_ld; double *curve_tol;         // SP_ANALYZE_SPLIT: [n_batch][curve_ld] lag curves (pp.ncurve lags + Y_PAD zeros), [n_batch] unc_tol
    int32_t *curve_list; long list_ld; int reach, cand_cap;  // ... and per frame [list_ld] int32: the filtered candidate count (-1: the frame went to the
                                                             // fallback list, -2: to far_list), then its candidates' lags as uint16 (scan_curve_kernel -> refine_list_kernel)
    int32_t *far_list;                                       // [0]: count, [4..]: batch-local indices of frames with a candidate whose PEAK lies beyond the lags
                                                             // refine_list_kernel holds (refine_far_kernel takes them with the whole curve)
};

// The last SP_TAIL lags of the curve.  The lag window falls below 1e-8 there (1e-10 .. 1e-17 over the last twelve lags), so
// the transforms' rounding error (~1e-16 S[0]) would be amplified into y by 1 / w_lag.  For an even frame length those
// entries only enter sinc sums under a taper that vanishes with them, but an odd n reads y[n - 1] directly
// (improve_extremum's ixmid >= nx arm, src/periodic.rs:194).  Those lags are sums of at most SP_TAIL products: lane l
// computes lag n - 1 - l in the reference's fold order (seed x[0], Q1) and overwrites the entry.
constexpr int SP_TAIL = 16;
__device__ __forceinline__ void spectral_exact_tail(double *ys, int n, const double *__restrict__ xf, const double *__restrict__ window,
                                                    const double *__restrict__ lag_window, double x0, double scale, int lane) {
    wave_sync();                                             // after the transform's own stores of these entries
    if (lane < SP_TAIL && lane < n) {
#pragma clang fp contract(off)
        const int L = n - 1 - lane;
        // Round 5: every sample the fold can need is requested before the first product (index 0 stands in past the lane's own
        // count) -- the loop `for i in 1..=lane` of rounds 2-4 waited for four loads per step, up to fifteen steps in a row:
        // 15 k cycles of a padded frame's 210 k in the kernel.  The same products added in the same order.
        double xu[SP_TAIL], wu[SP_TAIL], xv[SP_TAIL], wv[SP_TAIL];
#pragma unroll
        for (int i = 1; i < SP_TAIL; i++) {
            const int k = (i <= lane) ? i : 0;
            xu[i] = xf[k]; xv[i] = xf[L + k];
            wu[i] = (window != nullptr) ? window[k] : 1.0; wv[i] = (window != nullptr) ? window[L + k] : 1.0;
        }
        const double lw = lag_window[L];
        double acc = x0;
#pragma unroll
        for (int i = 1; i < SP_TAIL; i++) {
            const double u = (window != nullptr) ? xu[i] * wu[i] : xu[i];
            const double v = (window != nullptr) ? xv[i] * wv[i] : xv[i];
            const double next = acc + u * v;
            acc = (i <= lane) ? next : acc;
        }
        ys[L] = (acc * scale) / lw;
    }
}

// Levinson-Durbin on r[0..P] (src/spectrum.rs:63-84), every lane on the same (uniform) values
template <int P>
__device__ __forceinline__ void levinson_regs(const double (&r)[P + 1], double (&ac)[P + 1]) {
    double tmp[P + 1];
    double err = r[0];
    ac[0] = 1.0;
#pragma unroll
    for (int i = 1; i <= P; i++) ac[i] = 0.0;
#pragma unroll
    for (int i = 1; i <= P; i++) {
        double acc = r[i];
#pragma unroll
        for (int j = 1; j < i; j++) acc = acc + ac[j] * r[i - j];
        const double k = -acc / err;
        ac[i] = k;
#pragma unroll
        for (int j = 0; j < P; j++) tmp[j] = ac[j];
#pragma unroll
        for (int j = 1; j < i; j++) ac[j] = ac[j] + k * tmp[i - j];
        err = err * (1.0 - k * k);
    }
}

constexpr int SP_LPC_P = SPECTRAL_LPC_ORDER;

// LPC::lpc inside the fused call (round 6).  Rounds 1-5 ran the recursion in the frame's wavefront: sixty-four lanes on the same
// thirteen values, twelve IEEE divisions, ~450 vector instructions per frame for what ONE lane can do.  The kernels now store the
// lag sums r[0..12] in the frame's LPC row and levinson_rows_kernel_t (k_lpc.hip) turns the rows into coefficients afterwards, one
// row per LANE, in place -- the same operations in the same order (bit-identical rows), 64 x fewer instructions, and the
// conditioning probe + double-double redo of ill-conditioned rows (k_lpc_exact.hip) ride on that kernel for nothing.

// k_spectral_pow2.hip
int launch_analyze_pow2(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a);      // 1: ran in the split form

}  // namespace vbx
