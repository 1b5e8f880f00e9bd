// vbx_kernels.hpp -- host-side launcher declarations (one per kernel family).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define VBX_MAX_LPC_ORDER_K 62   // == VBX_MAX_LPC_ORDER (include/voxbox_hip.h)
#define VBX_MAX_FRAME_LEN_K 4096
#define VBX_MAX_RESONANCES_K 32
#define VBX_FORMANT_SLOTS_K 6

namespace vbx {

// A launch over a TIME SLICE of equal-length segments: item i is frame (i / tc) * seg_len + t0 + i % tc (skipped when
// it falls outside its segment or the batch).  seg_len == 0: the identity (item i is frame i).  Lets the sequential
// tracker start on the first frames of every utterance while the resonances of the later frames are still computed.
struct frame_map_t { long seg_len, t0, tc; };
__host__ __device__ inline long frame_map(const frame_map_t &m, long i, long n_frames) {
    if (m.seg_len == 0) return (i < n_frames) ? i : -1;
    const long t = m.t0 + i % m.tc, f = (i / m.tc) * m.seg_len + t;
    return (t < m.seg_len && t < m.t0 + m.tc && f < n_frames) ? f : -1;
}
inline long frame_map_items(const frame_map_t &m, long n_frames) {
    return m.seg_len == 0 ? n_frames : ((n_frames + m.seg_len - 1) / m.seg_len) * m.tc;
}


struct res_t { double frequency, bandwidth; };
struct pitch_t { double frequency, strength; };
struct cplx_t { double re, im; };
struct cplx32_t { float re, im; };

// k_lpc.hip
bool fewlags_supported(int n, int n_lags, bool want_lpc);
void launch_autocorr_fewlags(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                             int n_lags, int normalize, double *out_r, double *out_lpc, long lpc_ld = 0 /* 0: n_lags */,
                             int32_t *lpc_list = nullptr, int32_t *lpc_count = nullptr /* the conditioning probe's list (below) */);
void launch_autocorr_tiles(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                           int n_lags, double *out);
void launch_normalize_rows(hipStream_t s, double *data, long rows, int n);
void launch_levinson_rows(hipStream_t s, const double *r, long rows, long r_stride, int p, double *out, long out_ld,
                          double *out_kc = nullptr /* [rows, p] reflection coefficients */);
// The conditioning probe of the Levinson rows computed from FRAMES (round 6; reasoning at levinson_probe, vbx_spectral.hpp): the
// recursion is repeated on lag sums moved by +-LPC_PROBE_EPS of r[0]; a row that moves by more than LPC_PROBE_TOL in the parity metric
// (|a - b| / max(|b|, 1e-6 max |b|)) is listed and redone from the frame in double-double (k_lpc_exact.hip).
constexpr double LPC_PROBE_EPS = 16.0 * 2.220446049250313e-16;
#ifndef VBX_EXP_LPC_PROBE_TOL
#define VBX_EXP_LPC_PROBE_TOL 1e-6
#endif
constexpr double LPC_PROBE_TOL = VBX_EXP_LPC_PROBE_TOL;
void launch_levinson_rows_probe(hipStream_t s, const double *r, long rows, long r_stride, int p, double *out, long out_ld,
                                int32_t *lpc_list, int32_t *lpc_count, double *mfcc_rows = nullptr, long mfcc_ld = 0, int num_coeffs = 0,
                                const double *dct = nullptr /* p == 12: the same lane finishes the row's deferred MFCC tail */);

// k_burg.hip
bool burg_supported(int n, int p);
void launch_burg(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                 int p, double *out, int32_t *status, frame_map_t map = frame_map_t{0, 0, 0});
// the same on 16-bit PCM frames (samples widened in registers: s / 32767), f64 coefficients
void launch_burg_pcm16(hipStream_t s, const int16_t *x, long F, int n, long stride, const double *window,
                       int p, double *out, int32_t *status, frame_map_t map = frame_map_t{0, 0, 0});
void launch_burg_list(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                      int p, double *out, int32_t *status, const int32_t *list, const int32_t *count);
void launch_burg_pcm16_list(hipStream_t s, const int16_t *x, long F, int n, long stride, const double *window,
                            int p, double *out, int32_t *status, const int32_t *list, const int32_t *count);

// k_burg_fast.hip: the same coefficients from the frame's lag sums and edge samples (one pass over the frame), in chunks
// of burg_fast_chunk(F) items: launch_burg_lags, then launch_burg_recursion, per chunk; the frames its guard turns away
// are appended to burg_fast_list(ws, F, p) ([0] = count, indices from [2]) for launch_burg_list / launch_burg_pcm16_list.
// ws: burg_fast_scratch_bytes(F, p) bytes; the caller zeroes the list's count before the first chunk.
bool burg_fast_supported(int n, int p);
long burg_fast_chunk(long F);
size_t burg_fast_scratch_bytes(long F, int p);
int32_t *burg_fast_list(void *ws, long F, int p);
void launch_burg_lags(hipStream_t s, const double *x, long F, int n, long stride, const double *window, int p,
                      frame_map_t map, long i0, long m, void *ws);
void launch_burg_lags_pcm16(hipStream_t s, const int16_t *x, long F, int n, long stride, const double *window, int p,
                            frame_map_t map, long i0, long m, void *ws);
void launch_burg_recursion(hipStream_t s, long F, int p, frame_map_t map, long i0, long m, double *out, int32_t *status, void *ws);

void launch_count_accumulate(hipStream_t s, const int32_t *count, int32_t *total, bool reset);   // total = (reset ? 0 : total) + count

// k_roots.hip
void launch_find_roots(hipStream_t s, cplx_t *polys, long F, int len, int32_t *status);
void launch_laguerre(hipStream_t s, const cplx_t *polys, long F, int len, cplx_t start, cplx_t *out);
void launch_div_polynomial(hipStream_t s, cplx_t *polys, const cplx_t *others, long F, int len, cplx_t *rem, int32_t *status);
void launch_find_roots_f32(hipStream_t s, cplx32_t *polys, long F, int len, int32_t *status);
void launch_laguerre_f32(hipStream_t s, const cplx32_t *polys, long F, int len, cplx32_t start, cplx32_t *out);
void launch_to_resonance(hipStream_t s, const cplx_t *roots, long F, int n_roots, double sample_rate,
                         int strict_im, res_t *out, int out_stride, int32_t *out_count, const int32_t *status);
// Burg coefficients [F,p] -> reversed complex polynomial -> roots -> resonances [F,32] (find_formants core)
void launch_formant_resonances(hipStream_t s, const double *coeffs, long F, int p, double sample_rate,
                               res_t *out_res, int32_t *out_count, int32_t *status, frame_map_t map = frame_map_t{0, 0, 0});
// k_roots_fast.hip: the same rows from the real polynomial's conjugate pairs (converged Laguerre solves, quadratic
// deflation, a polish step that doubles as the check); a frame that fails the check is done again inside the kernel by
// the reference's iteration.  redo_count: optional device counter of those frames.
bool formant_resonances_fast_supported(int p);
void launch_formant_resonances_fast(hipStream_t s, const double *coeffs, long F, int p, double sample_rate,
                                    res_t *out_res, int32_t *out_count, int32_t *status, frame_map_t map, int32_t *redo_count);

// k_tracker.hip
void launch_tracker(hipStream_t s, const res_t *res, long F, int n_res, const int32_t *res_count,
                    const int64_t *seg_start, long n_seg, const res_t *est_init, int n_est,
                    const int32_t *frame_status, res_t *out, long out_ld /* doubles per output row, >= 2*n_est */,
                    long t0 = 0 /* first frame of every segment's slice */, long tc = 0x7fffffffffffffffL /* frames per slice */);

// the same scan for batches with long utterances: speculative chunks + exact repair (k_tracker.hip)
size_t tracker_chunked_workspace_bytes(long F);
void launch_tracker_chunked(hipStream_t s, const res_t *res, long F, int n_res, const int32_t *res_count,
                            const int64_t *seg_start, long n_seg, const res_t *est_init, int n_est,
                            const int32_t *frame_status, res_t *out, long out_ld, void *ws);

// a shard's track continued from the row the previous shard ends with (k_tracker.hip, tracker_stitch_kernel)
void launch_tracker_stitch(hipStream_t s, const res_t *res, long F, int n_res, const int32_t *res_count, int n_est,
                           const int32_t *frame_status, res_t *out, long out_ld, long first, long stop,
                           const double *state_in /* 2 n_est doubles */, int32_t *changed /* rows rewritten, or NULL */);

// k_pitch.hip
size_t pitch_lds_bytes(int n);
// the sorted candidate list lives one entry per lane up to this many entries; a larger kmax parks the whole
// candidate Vec in an extra LDS region of pitch_full_list_bytes(n, kmax) and rank-sorts it (nothing is pruned)
constexpr int PITCH_LIST_LANES = 64;
size_t pitch_full_list_bytes(int n, int kmax);
// profiling counters of the refine kernel: [PITCH_WORK_SLOTS][4] = frames, candidates, sinc evaluations, sinc terms
constexpr int PITCH_WORK_SLOTS = 64;
// experiment builds (-DVBX_EXP_PHASES, tools/experiments/phases.sh): behind the counters, [PITCH_WORK_SLOTS][PHASE_SLOTS] sums of
// shader-clock cycles a wavefront spent in each phase of the fused kernel (s_memtime at the phase boundaries, after a
// wait for everything outstanding).  A CU holds a fixed number of frames (LDS), so frames/s = frames in flight / a frame's
// time in the kernel: the phase that holds a wavefront longest is the one to shorten, whatever its instruction count.
constexpr int PHASE_SLOTS = 16;
constexpr int PITCH_WORK_WORDS = PITCH_WORK_SLOTS * 4 + PITCH_WORK_SLOTS * PHASE_SLOTS;
#ifdef VBX_EXP_PHASES
#define VBX_PHASE_INIT() unsigned long long _tp = __builtin_readcyclecounter()
#define VBX_PHASE(work_, f_, k_)                                                                                   \
    do {                                                                                                            \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                 \
        const unsigned long long _t = __builtin_readcyclecounter();                                                 \
        if ((work_) != nullptr && lane_id() == 0)                                                                   \
            atomicAdd((work_) + PITCH_WORK_SLOTS * 4 + ((f_) & (PITCH_WORK_SLOTS - 1)) * PHASE_SLOTS + (k_), _t - _tp); \
        _tp = _t;                                                                                                   \
    } while (0)
#else
#define VBX_PHASE_INIT() do { } while (0)
#define VBX_PHASE(work_, f_, k_) do { } while (0)
#endif
void launch_pitch(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                  const double *lag_window, double sample_rate, double threshold, double fmin, double fmax,
                  int kmax, pitch_t *out_cand, long cand_ld /* doubles per output row, >= 2*kmax, even */,
                  int32_t *out_count, int32_t *status, unsigned long long *work);
// the same kernel over a device-resident list of frame indices (fallback of k_spectral.hip), fixed grid
void launch_pitch_list(hipStream_t s, const int32_t *frame_list, const int32_t *list_count, int grid,
                       const double *x, int n, long stride, const double *window,
                       const double *lag_window, double sample_rate, double threshold, double fmin, double fmax,
                       int kmax, pitch_t *out_cand, long cand_ld, int32_t *out_count, int32_t *status,
                       unsigned long long *work, bool pcm = false /* x points to int16 PCM samples */);
void launch_sinc_points(hipStream_t s, const double *y, int ylen, long offset, long nx, const double *xs, long m,
                        long depth, double *out, int32_t *status);
void launch_extremum_points(hipStream_t s, const double *y, int ylen, long offset, long nx, const double *ix, long m,
                            long depth, double *out_xy, int32_t *status, int interp = 2 /* 0 None, 1 Parabolic, 2 Sinc */,
                            int is_max = 1);

// k_spectral.hip, k_spectral_pow2.hip: pitch + LPC + MFCC from one real FFT of the zero-padded frame.  Three transform
// sizes ("plans"), named by the complex FFT length Nc (the real transform has 2 Nc points, the frame at most Nc samples):
// 1200 (25 ms at 48 kHz, and 1025..1199, e.g. 25 ms at 44.1 kHz), 1024 (512..1024), 2048 (1201..2048) and 4096 (2049..4096):
// the reference's own test and example shapes (tests/lib.rs:56, examples/pitch_detection.rs:23) and everything between.
constexpr int SPECTRAL_N = 1200;
constexpr int SPECTRAL_LPC_ORDER = 12;
constexpr int SPECTRAL_TAB_COMPLEX = 60 * 20 + 3 * 20 + 601;    // W_1200^(n' ka) | W_60^(c kb) | W_2400^m
enum { SPECTRAL_PLAN_NONE = 0, SPECTRAL_PLAN_1200 = 1, SPECTRAL_PLAN_1024 = 2, SPECTRAL_PLAN_2048 = 3, SPECTRAL_PLAN_4096 = 4,
       SPECTRAL_PLANS = 5 };
constexpr int SPECTRAL_AC_MIN_LAGS = 64;                        // vbx_autocorrelate_f64 below 1024 samples: fewer lags -> the direct kernel
                                                                // (from 1024 samples on the FFT wins wherever the few-lag kernel does not apply)
constexpr int SPECTRAL_MIN_N = 512;                             // shorter frames: the direct lag sums are as fast (measured)
int spectral_plan(int n);                                       // the plan that serves frame length n, or SPECTRAL_PLAN_NONE
int spectral_plan_mfcc(int n);                                  // the same when MFCC should join the fused kernel (k_spectral.hip)
bool spectral_supported_plan(int plan, int n, int lpc_order, int mfcc_nb, int mfcc_b_lo, int num_coeffs);
int spectral_plan_nc(int plan);                                 // its complex FFT length
int spectral_tab_complex(int plan);                             // complex entries of its twiddle table
void spectral_fill_tab(int plan, double *h_out);                // the table, host side: [2 * spectral_tab_complex(plan)]
int spectral_pow2_tab_complex(int plan);
void spectral_pow2_fill_tab(int plan, double *h_out);
// MFCC::mfcc inside the fused kernel at a frame length n that does not divide the transform's M (src/spectrum.rs:401-441 wants
// the n-point DFT: bin k is the frame's DTFT at k / n, between the transform's bins).  The frame occupies n of the transform's M >= 2 n
// samples, so its DTFT is a band-limited function of the bin index sampled at more than twice its rate: with c = (n - 1) / 2
//     X(k / n) e^{i w c} = sum_j  Z[j] K0(k M / n - j),      Z[j] = X_M[j] e^{2 pi i j c / M},
// exactly, for any real K0 whose transform is 1 on |t| <= n / 2M and 0 on |t - m| <= n / 2M, m != 0.  K0 = sinc * (the transform of a
// Kaiser-Bessel bump of half-width 1/2 - n / 2M) cut to 24 / 32 / 40 taps by M / n: the cut's error is < 1e-14 of the largest |X| of
// the transform at every M / n >= 2 (tests/test_mfcc_interp_table.py holds the tables to the exact DFT on the CPU) -- MFCC values
// within 1e-10 of the chirp-z kernel's on speech.  The phase factor drops out of |X|^2.  ~600 vector instructions per frame instead of a chirp-z kernel of two more
// transforms.  Host tables: rot[j] = e^{2 pi i j c / M} (j <= M / 4), per bin its first tap's index and its taps.
constexpr int MFCC_INTERP_MAX_TAPS = 40;                        // 24, 32 or 40 taps per bin (mfcc_interp_taps: by M / n)
struct mfcc_interp_t {
    const double *rot;                                          // complex [M / 4 + 1]
    const double *coef;                                         // double2 [slots][TAPS / 2][NT]: taps 2 t, 2 t + 1 of bin NT u + thread (NT threads per frame,
                                                                // slots = ceil(nb / NT))
    const int32_t *j0;                                          // [slots][NT]: index of the bin's first tap MINUS jmin
    int jmin, jmax;                                             // the transform's bins the taps read: Z[jmin .. jmax] (jmin may be < 0)
    int taps;                                                   // per bin (a multiple of 8)
    int pu_off;                                                 // doubles from the exchange buffer's start to the filter products
    int lds_bytes;                                              // Z + products + mel sums
};
int mfcc_interp_taps(int plan, int n);
size_t mfcc_interp_table_bytes(int plan, int nb);
size_t mfcc_interp_coef_offset(int plan);                      // byte offsets of coef and j0 in the table (rot at 0)
size_t mfcc_interp_j0_offset(int plan, int nb);
// false: the shape has no interpolated form (bins beyond M / 4, LDS that would cost a frame per CU)
bool mfcc_interp_fill(int plan, int n, int b_lo, int nb, void *h_table, mfcc_interp_t *h_desc /* offsets in the pointer fields */);

struct spectral_launch_t {
    int plan; int n;                                             // spectral_plan(n), frame length
    const double *x; long F; long stride; const double *window; const double *lag_window; const double *tab;
    bool mfcc_defer;                                             // the MFCC rows leave the kernel as filter sums; launch_mfcc_rows finishes them (num_coeffs <= 16)
    bool lag_rcp;                                                // lag_window[((n + 1) & ~1) + i] = RN(1 / lag_window[i]) (quotient_by_table, vbx_spectral.hpp)
    double sample_rate, threshold, fmin, fmax; int kmax;
    pitch_t *out_cand; long cand_ld; int32_t *out_count; int32_t *pitch_status; unsigned long long *work;
    double *out_lpc; long lpc_ld;                                // NULL: no LPC
    double *out_mfcc; long mfcc_ld; int32_t *mfcc_status;        // NULL: no MFCC
    const int32_t *bins; const double *slopes; const double *dct; int num_coeffs; int nb;
    int32_t *unsure_list; int32_t *unsure_count;                 // frames handed to launch_pitch_list
    int32_t *lpc_list; int32_t *lpc_count;                       // frames handed to launch_lpc_exact_list (ill-conditioned LPC rows); NULL: no probe
    bool mfcc_only;                                              // MFCC::mfcc alone (n == the plan's Nc): no pitch, no LPC
    double *out_r; int n_lags;                                   // non-NULL: Autocorrelate::autocorrelate(n_lags) alone, [F, n_lags]
    bool pcm;                                                    // x points to int16 PCM samples (n == 1200 only)
    bool whole_curve;                                            // keep every lag of the curve in LDS (VBX_PITCH_CURVE_CUT=0; tests)
    bool interp; mfcc_interp_t ip;                               // MFCC by interpolation of the transform's bins (device pointers)
    double *curve_ws; size_t curve_ws_bytes;                     // scratch for the split form of the 4096-point plan (lag curves between its two kernels); NULL: fused
};
size_t spectral_split_row_bytes(int n, double sample_rate, double fmin);   // bytes per frame of that scratch, 0: the shape has no split form
bool spectral_supported(int n, int lpc_order, int mfcc_nb, int mfcc_b_lo, int num_coeffs);
int launch_analyze(hipStream_t s, const spectral_launch_t &L);       // 1: the call ran as two kernels (SP_ANALYZE_SPLIT), 0: one
// k_lpc_exact.hip: LPC::lpc(p) of the listed frames from double-double lag sums and a double-double recursion (the exact
// answer rounded once), over a list only the device knows the length of
// k_mfcc.hip: log10 (clamped at 1e-10) + DCT of rows of mel filter sums, in place, a lane per row (the deferred tail of MFCC::mfcc, vbx_mfcc_tail.hpp)
void launch_mfcc_rows(hipStream_t s, double *rows, long F, long ld, int num_coeffs, const double *dct);
bool lpc_exact_supported(int n, int p);                               // frames of 2..4096 samples, orders 1..31
void launch_lpc_exact_list(hipStream_t s, const int32_t *frame_list, const int32_t *list_count, int cus, const double *x, int n,
                           long stride, const double *window, bool pcm, int p, double *out_lpc, long lpc_ld);

// the f32 instantiation (Sample = f32): the same kernels with float frames and float outputs
void launch_autocorr_fewlags_f32(hipStream_t s, const float *x, long F, int n, long stride, const float *window,
                                 int n_lags, int normalize, float *out_r, float *out_lpc, long lpc_ld = 0);
void launch_autocorr_tiles_f32(hipStream_t s, const float *x, long F, int n, long stride, const float *window,
                               int n_lags, float *out);
void launch_normalize_rows_f32(hipStream_t s, float *data, long rows, int n);
void launch_levinson_rows_f32(hipStream_t s, const float *r, long rows, long r_stride, int p, float *out, long out_ld,
                              float *out_kc = nullptr);
void launch_burg_f32(hipStream_t s, const float *x, long F, int n, long stride, const float *window,
                     int p, float *out, int32_t *status);
void launch_widen_frames(hipStream_t s, const float *x, long F, int n, long stride, const float *window, double *out);
// k_f32.hip: the reference-faithful f32 forms (every fold in f32, in source order; bit-identical to the f32 restatement the tests hold them to)
void launch_autocorr_f32_exact(hipStream_t s, const float *x, long F, int n, long stride, const float *window, int n_lags, float *out);
size_t pitch_f32_exact_lds_bytes(int n, int kmax);
void launch_pitch_f32_exact(hipStream_t s, const float *x, long F, int n, long stride, const float *window, const float *lag_window32,
                            double sample_rate, double threshold, double fmin, double fmax, int kmax, pitch_t *out_cand, long cand_ld,
                            int32_t *out_count, int32_t *status);
void launch_levinson_f32_exact(hipStream_t s, const float *r, long rows, long r_stride, int p, float *out, long out_ld, float *out_kc);
size_t burg_f32_exact_scratch_bytes(long frames, int n);
void launch_burg_f32_exact(hipStream_t s, const float *x, long f0, long f1, long F, int n, long stride, const float *window, int p,
                           float *out, int32_t *status, float *scratch);
void launch_narrow(hipStream_t s, const double *in, long count, float *out);

// k_mfcc.hip
bool mfcc_fits(int n, int nb);
struct mfcc_plan_t { bool ok; int n1, n2, nc, tm; };     // two-stage DFT geometry: n = n1*n2, nc = padded stage-1 columns
mfcc_plan_t mfcc_plan(int n, int nb);
void launch_mfcc_dft2(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                      const mfcc_plan_t &pl, const double *ctab /* [n1][nc] */, const double *twid /* [n][2] */,
                      const int32_t *bins_dev, const double *slopes, const double *dct_table, int num_coeffs, double *out,
                      long out_ld /* doubles per output row */, int32_t *status /* set to 0 per frame, or NULL */, int nb, int cu_count);
// k_mfcc_mfma.hip: both DFT stages on the matrix cores
struct mfcc_mplan_t { bool ok; int n1, n2, k2, mt, ntd, ntm, src0, src1; };
mfcc_mplan_t mfcc_mfma_plan(int n, int b_lo, int nb);
void launch_mfcc_mfma(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                      const mfcc_mplan_t &pl, const double *ctab, const double *twd, const double *twm, const double *wm,
                      const int32_t *bins_dev, const double *slopes, const double *dct_table, int num_coeffs, double *out,
                      long out_ld /* doubles per output row */, int32_t *status /* set to 0 per frame, or NULL */, int nb, int cu_count);
void launch_mfcc(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                 const double *kappa_sigma /* [nb][2] Goertzel-Reinsch constants */, const int32_t *bins /* K+2 */,
                 const double *slopes /* [nb][2] */, const double *dct_table /* [K][K] */,
                 int num_coeffs, double *out, long out_ld, int32_t *status, int nb /* bins[K+1]-bins[0] */);
// k_mfcc_czt.hip: the needed bins of the n-point DFT by the chirp-z identity on the power-of-two FFT (any n with
// n + top - 1 <= 4096: frame lengths that are not a transform's and do not divide one; longer ones up to 4096 samples split
// into two blocks of n1 samples whose complex results add up, n1 + top - 1 <= 4096)
int mfcc_czt_plan(int n, int top /* b_lo + nb */);              // SPECTRAL_PLAN_1024 / _2048 / _4096, or SPECTRAL_PLAN_NONE
int mfcc_czt_split_plan(int n, int top, int *n1);               // the same for a frame split in two (n1 = samples per block)
// n1 <= 0 or >= n: one block, h_bhat [2 L]; else ceil(n / n1) blocks, h_bhat [blocks][2 L]
void mfcc_czt_fill_tabs(int n, int top, int L, int n1, double *h_chirp /* [2 n] */, double *h_bhat);
void launch_mfcc_czt(hipStream_t s, int plan, const double *x, long F, int n, int n1, long stride, const double *window,
                     const double *tab /* the plan's twiddles */, const double *chirp, const double *bhat, const int32_t *bins,
                     const double *slopes, const double *dct, int num_coeffs, int nb, double *out, long out_ld, int32_t *status,
                     double *cw_scratch /* [blocks][2 L] */);
void launch_fill_rows(hipStream_t s, double *out, long rows, int n, long ld, double value, int32_t *status, int32_t code);
void launch_dct_rows(hipStream_t s, const double *in, long rows, int n, const double *dct_table, double *out);

// k_front.hip
void launch_resample(hipStream_t s, const double *x, long F, int n, long stride, const int32_t *tab_idx,
                     const double *tab_frac, int m, double *out);
void launch_ring_frames(hipStream_t s, const double *ring, long capacity, long head, long F, int n, long stride, double *out);
void launch_pcm16(hipStream_t s, const int16_t *pcm, size_t n, double denom, double *out);
void launch_rms(hipStream_t s, const double *x, long F, int n, long stride, const double *window, double *out);
void launch_preemphasis(hipStream_t s, const double *x, long F, int n, long stride, double c, double *out);

// k_long.hip: frames of more than VBX_MAX_FRAME_LEN_K samples (tiles out of HBM / L2 instead of registers / LDS)
size_t burg_long_scratch_bytes(long frames, long n);
void launch_burg_long(hipStream_t s, const double *x, long f0, long f1, long F, long n, long stride, const double *window,
                      int p, double *out, int32_t *status, double *ws /* burg_long_scratch_bytes(f1 - f0, n) */);
size_t autocorr_long_scratch_bytes(long F, long n, long n_lags);
void launch_autocorr_long(hipStream_t s, const double *x, long F, long n, long stride, const double *window, long n_lags,
                          double *out, double *ws);
size_t pitch_long_scratch_bytes(long F, long n);
double *pitch_long_r(void *ws);                                  // [F][n]: the caller fills it with the NORMALISED lag sums of every frame
void launch_pitch_long(hipStream_t s, long F, long n, const double *lag_window, double sample_rate, double threshold, double fmin,
                       double fmax, int kmax, double *out_cand, long cand_ld, int32_t *out_count, int32_t *status, void *ws);
size_t mfcc_long_scratch_bytes(long F, int nb);
void launch_mfcc_long(hipStream_t s, const double *x, long F, long n, long stride, const double *window, const double *kappa_sigma,
                      const int32_t *bins, const double *slopes, const double *dct, int num_coeffs, int nb, double *out, long out_ld,
                      int32_t *status, double *ws);
size_t preemphasis_long_scratch_bytes(long F, long n);
void launch_preemphasis_long(hipStream_t s, const double *x, long F, long n, long stride, double c, double *out, double *ws);

// k_synth.hip
void launch_synth(hipStream_t s, double *out, size_t n_samples, uint64_t sample_offset, double sample_rate, uint64_t seed);

// k_selftest.hip
void launch_selftest(hipStream_t s, double *out /* 64*8 */);

}  // namespace vbx
