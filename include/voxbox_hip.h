/*
 * voxbox_hip.h -- C ABI of libvoxbox_hip.so: batched, MI355X-native (gfx950)
 * replacement for the per-frame DSP hot path of the Rust crate vox_box 0.3.0.
 *
 * The reference exposes this path as extension traits on slices, called once
 * per frame from a user loop (examples/pitch_detection.rs:23-30,
 * tests/lib.rs:71-83).  It has no FFI of its own, so these
 * entry points ARE what a Rust/cgo/ctypes binding of the path would bind:
 * one call = the user's whole frame loop (F frames).  Each declaration cites
 * the reference interface it replaces (file:line under /root/reference).
 *
 * Conventions
 *  - plain pointers and sizes only; every data pointer is a DEVICE pointer
 *    unless the parameter name starts with `h_` (host).  vbx_malloc/vbx_memcpy_*
 *    are provided so a caller needs no HIP binding of its own; pointers from
 *    hipMalloc / torch tensors (.data_ptr()) are equally valid.
 *  - a frame batch is (x, n_frames, frame_len, stride, window): frame t is
 *    x[t*stride .. t*stride+frame_len) -- stride == frame_len is the dense
 *    [F, N] batch, stride == hop is the Windower view into contiguous audio
 *    (sample 0.10 Windower: frame t exists while frame_len <= remaining).
 *    `window` (device, frame_len doubles, or NULL) is multiplied onto the
 *    samples on load: the batched form of window::Windower::hanning, whose
 *    frames the reference's traits receive already windowed.
 *  - all arithmetic is f64 ("Sample = f64" instantiation of the traits).  The f32 instantiation (vbx_*_f32: float
 *    frames in, float results out; vbx_*_c32: Complex<f32> Polynomial) widens on load, computes in f64 and rounds
 *    each result to f32 once -- except the Complex<f32> root finder, which follows the reference's f32 arithmetic
 *    step by step because its iteration counts and root ORDER depend on it.
 *  - calls are asynchronous on the context's HIP stream; vbx_sync() waits.
 *  - a context (stream, cached tables, scratch) is not internally synchronised: one host thread per context at
 *    a time.  Contexts are independent of each other and cheap; use one per thread / stream.
 *  - return value: 0 ok, <0 API misuse / runtime failure (vbx_last_error()).
 *    Per-frame conditions that make the reference return Err or panic are
 *    reported in an int32 status[F] array (codes below) so one bad frame never
 *    aborts a batch; outputs of such a frame are zero-filled and tracker state
 *    passes through unchanged (src/lib.rs:75 `?`).
 *  - there is NO CPU fallback: without a HIP device vbx_ctx_create fails.
 */
#ifndef VOXBOX_HIP_H
#define VOXBOX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBX_ABI_VERSION 5

/* API return codes */
#define VBX_SUCCESS 0
#define VBX_E_INVALID (-1)      /* null pointer / bad size / unsupported shape */
#define VBX_E_RUNTIME (-2)      /* HIP runtime error */
#define VBX_E_NODEVICE (-3)     /* no usable gfx950 device */

/* per-frame status codes (VoxBoxError, src/error.rs:4-16, and the panics of the path) */
#define VBX_FRAME_OK 0
#define VBX_FRAME_ERR_LPC 1         /* Err(LPC("Denum was <= 0.0")), src/spectrum.rs:123-125 */
#define VBX_FRAME_ERR_POLYNOMIAL 2  /* Err(Polynomial(..)), src/polynomial.rs:95,123 */
#define VBX_FRAME_ERR_NAN 3         /* partial_cmp().unwrap() on NaN, src/periodic.rs:453 */
#define VBX_FRAME_ERR_PANIC 4       /* any other panic of the reference (index out of bounds, assert) */

#define VBX_MAX_RESONANCES 32       /* MAX_RESONANCES, src/lib.rs:26 */
#define VBX_FORMANT_SLOTS 6         /* FormantSlots, src/spectrum.rs:228 */
#define VBX_MAX_LPC_ORDER 62        /* order of lpc / lpc_praat / find_formants: its at most order / 2 resonances fit the reference's
                                       MAX_RESONANCES, its polynomial (order + 1 coefficients) the root finder's VBX_MAX_POLY_LEN */
#define VBX_MAX_POLY_LEN 64         /* coefficients of a polynomial of vbx_find_roots_* / vbx_laguerre_* / vbx_div_polynomial_* */
#define VBX_MAX_FRAME_LEN 4096      /* frames up to here live in registers / LDS (the fast kernels); the f32 instantiation
                                       (vbx_*_f32, vbx_*_f32_wide) takes no longer ones */
#define VBX_MAX_LONG_FRAME_LEN 67108864 /* 2^26: every f64 frame-batch entry point takes frames up to this length -- the reference's
                                       slices have no cap (tests/lib.rs:27-41 passes a 31,232-sample file as ONE frame of
                                       find_formants).  Beyond VBX_MAX_FRAME_LEN the kernels walk the frame in tiles out of HBM
                                       (k_long.hip): exact, but built for a whole recording per frame, not for millions of them;
                                       vbx_pitch_f64 keeps the frame's 2 * frame_len lag curve in HBM and needs frame_len < 2^30,
                                       and takes kmax up to VBX_PITCH_MAX_CANDIDATES(frame_len) there */
#define VBX_MAX_PITCH_CANDIDATES 1026 /* kmax upper bound of vbx_pitch_f64: frame_len/4 strict local maxima in
                                        [0, frame_len/2) + the unvoiced candidate, at VBX_MAX_FRAME_LEN */
/* the whole Vec of a frame never has more than this many entries (out_count <= vbx_pitch_max_candidates) */
#define VBX_PITCH_MAX_CANDIDATES(frame_len) ((frame_len) / 4 + 2)

typedef struct vbx_ctx vbx_ctx;

/* #[repr(C)] Resonance<f64>, src/spectrum.rs:149-154 */
typedef struct { double frequency; double bandwidth; } vbx_resonance;
/* Pitch<f64>, src/periodic.rs:306-310 */
typedef struct { double frequency; double strength; } vbx_pitch;
/* num::Complex<f64> (repr(C): re, im) */
typedef struct { double re; double im; } vbx_complex;

/* MALE/FEMALE_FORMANT_ESTIMATES, src/lib.rs:27-28 */
extern const double VBX_MALE_FORMANT_ESTIMATES[4];
extern const double VBX_FEMALE_FORMANT_ESTIMATES[4];

/* ------------------------------------------------------------------ context */

int vbx_abi_version(void);
/* device: HIP device ordinal.  hip_stream: a hipStream_t to launch on (e.g. torch's
 * current stream), or NULL to let the context create and own one. */
int vbx_ctx_create(vbx_ctx **out, int device, void *hip_stream);
void vbx_ctx_destroy(vbx_ctx *ctx);
int vbx_sync(vbx_ctx *ctx);
const char *vbx_last_error(const vbx_ctx *ctx); /* ctx may be NULL: last global error */
/* device name (e.g. "gfx950:sramecc+:xnack-"), CU count; any pointer may be NULL */
int vbx_device_info(const vbx_ctx *ctx, char *h_name, size_t name_cap, int *h_cu_count);

/* device memory helpers (synchronous w.r.t. the host for memcpy) */
int vbx_malloc(vbx_ctx *ctx, void **out_dptr, size_t bytes);
int vbx_free(vbx_ctx *ctx, void *dptr);
int vbx_memcpy_h2d(vbx_ctx *ctx, void *dst, const void *h_src, size_t bytes);
int vbx_memcpy_d2h(vbx_ctx *ctx, void *h_dst, const void *src, size_t bytes);
int vbx_memset(vbx_ctx *ctx, void *dst, int value, size_t bytes);

/* HIP-event timing on the context's stream (bench.py's roofline leg).
 * vbx_timer_begin/end bracket a region; *h_ms is valid after the call returns. */
int vbx_timer_begin(vbx_ctx *ctx);
int vbx_timer_end(vbx_ctx *ctx, float *h_ms);
/* Per-kernel event profile: when enabled every kernel launch is bracketed by
 * events; vbx_profile_get sums them by kernel name (synchronises the stream). */
int vbx_profile_enable(vbx_ctx *ctx, int on);
int vbx_profile_reset(vbx_ctx *ctx);
int vbx_profile_get(vbx_ctx *ctx, const char *kernel_name, double *h_total_ms, long *h_launches);
/* ABI 5.  The stream the kernel's last profiled launch ran on: 0 = the context's stream (the critical path of a call),
 * 1 = the side stream of the fused frame loop (the formant chain / an unfused MFCC beside the spectral kernel),
 * 2 = the tracker's time-slice stream; -1 = not profiled.  Event times of kernels on streams 1 and 2 include the time
 * they spend co-resident with the context stream's kernel: a bench must not call them "dominant" by that number. */
int vbx_profile_stream(vbx_ctx *ctx, const char *kernel_name, int *h_stream);
/* Work the pitch refine kernel executed while profiling was enabled (since the last
 * vbx_profile_reset): h_out4 = { frames, candidates found, sinc evaluations, sinc terms }.
 * Feeds bench.py's FP64 roofline with the work actually done, not the reference's. */
int vbx_profile_pitch_work(vbx_ctx *ctx, uint64_t *h_out4);
/* names of profiled kernels, '\n'-separated, into h_buf */
int vbx_profile_names(vbx_ctx *ctx, char *h_buf, size_t cap);

/* ------------------------------------------------------------------ tables (host) */

/* sample 0.10 window tables, built on the host with the reference's recurrences.
 *  VBX_WINDOW_HANNING          Window::<Hanning>::new(n): phase accumulated by 1/(n-1), % 1.0
 *                              (Windower::hanning, examples/pitch_detection.rs:23)
 *  VBX_WINDOW_HANNING_LAG      HanningLag::at_phase over the same phases (src/periodic.rs:236-248,:400)
 *  VBX_WINDOW_HANNING_PERIODIC Hanning::at_phase(idx/len) (src/lib.rs:66-70)
 *  VBX_WINDOW_RECTANGLE        all ones (Windower::rectangle, tests/lib.rs:71) */
#define VBX_WINDOW_HANNING 0
#define VBX_WINDOW_HANNING_LAG 1
#define VBX_WINDOW_HANNING_PERIODIC 2
#define VBX_WINDOW_RECTANGLE 3
int vbx_window_table_f64(int kind, size_t n, double *h_out);
/* number of Windower frames: (n_samples - frame_len)/hop + 1 while frame_len <= remaining */
size_t vbx_frame_count(size_t n_samples, size_t frame_len, size_t hop);

/* hz_to_mel / mel_to_hz, src/spectrum.rs:375-381 (host scalars) */
double vbx_hz_to_mel(double hz);
double vbx_mel_to_hz(double mel);
/* find_formants_real_work_size / _complex_work_size, src/lib.rs:30-36.  The library owns
 * its workspaces; these exist so ported callers that size buffers keep compiling. */
size_t vbx_find_formants_real_work_size(size_t buf_len, size_t n_coeffs);
size_t vbx_find_formants_complex_work_size(size_t n_coeffs);

/* ------------------------------------------------------------------ periodic.rs */

/* Autocorrelate::autocorrelate(n_lags) per frame (src/periodic.rs:265-289), including the
 * fold seed quirk r[lag] = x[0] + sum_{i>=1} x[i]*x[i+lag].  out: [F, n_lags]. */
int vbx_autocorrelate_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len,
                          size_t stride, const double *window, size_t n_lags, double *out);

/* Normalize::normalize on each row (src/waves.rs:60-76): row *= 1/max|row|.  data: [F, n] in place. */
int vbx_normalize_f64(vbx_ctx *ctx, double *data, size_t n_rows, size_t n);

/* interpolate_sinc (src/periodic.rs:29-87) of one lag curve y[ylen] at M query points.
 * status[M] (optional) receives VBX_FRAME_ERR_PANIC where the reference would index out of bounds. */
int vbx_interpolate_sinc_f64(vbx_ctx *ctx, const double *y, size_t ylen, long offset, size_t nx,
                             const double *xs, size_t m, size_t max_depth, double *out, int32_t *status);

/* improve_extremum(.., Interpolation::Sinc(depth), is_max = true) (src/periodic.rs:192-229)
 * at M starting points; out_xy: [M, 2] = (xmid, ymid). */
int vbx_improve_extremum_f64(vbx_ctx *ctx, const double *y, size_t ylen, long offset, size_t nx,
                             const double *ixmid, size_t m, size_t depth, double *out_xy, int32_t *status);
/* improve_extremum with every arm of `Interpolation` (src/periodic.rs:89-93,192-229) and the is_max flag: NONE returns
 * (0, y[0]) (:197-199), PARABOLIC the three-point fit around floor(ixmid) (:200-207; status PANIC where the reference indexes
 * out of bounds), SINC(depth) the Brent search of the interpolant -- negated by the closure when is_max == 0 (:219-222).
 * Only SINC with is_max != 0 is on the pitch path; the rest is the crate's public surface. */
#define VBX_INTERP_NONE 0
#define VBX_INTERP_PARABOLIC 1
#define VBX_INTERP_SINC 2
int vbx_improve_extremum_ex_f64(vbx_ctx *ctx, const double *y, size_t ylen, long offset, size_t nx,
                                const double *ixmid, size_t m, int interpolation, size_t depth, int is_max,
                                double *out_xy, int32_t *status);

/* Pitched::pitch::<Hanning>(sample_rate, threshold, _, _, min, max) per frame
 * (src/periodic.rs:356-358,396-455; local_peak/global_peak are unused by the reference).
 * out_cand: [F, kmax] candidates, stable-sorted by descending strength exactly as the
 * reference's Vec (entries past count are zero); out_count[F] = full candidate count
 * (may exceed kmax).  PitchExtractor (src/periodic.rs:337-353) output = out_cand[f*kmax + 0].
 * Only the kmax entries that are returned are guaranteed to have been refined: a candidate whose strength
 * is provably below the kmax-th best is skipped (exact -- the returned entries, the count and the status
 * are the reference's; DESIGN.md "exact top-k pruning").  kmax = 1 is the fast path.
 * The WHOLE Vec of every frame (src/periodic.rs:452-454 returns all of it) is retrievable: either
 * kmax = VBX_PITCH_MAX_CANDIDATES(frame_len), which no frame can exceed, or the two-call protocol -- a first call
 * with kmax = 1 yields out_count[F], a second call with kmax = max(out_count) returns every entry.  kmax > 64
 * switches the kernel from its lane-resident list to an LDS-resident one (nothing is pruned, every candidate is
 * refined as in the reference; slower).
 * Bit identity across kmax: the lists returned for kmax in {1, 2, 3} are bit for bit the head of one another, and so are
 * the lists for every kmax >= 4 (from 4 on, few-candidate frames refine four candidates at a time, which changes the last
 * bits of a candidate's sinc sums); between the two classes a candidate agrees to ~1e-7 relative in Hz.
 * Which shapes are fast (one MI355X, kmax = 1, frames/s; the table in DESIGN.md section 4 is kept current): 512..1024 samples
 * 44-56 M, 1025..1200 39-42 M, 1201..2048 29 M, 2049..4096 16-17 M (that range runs its refinement in kernels of its
 * own, with the lag curves in a context-owned scratch buffer between them: <= 2.3 GB, 4.4 GB at an odd length),
 * below 512 samples the direct lag sums on the matrix cores 50-70 M.  kmax 2 / 8 / 64 / whole Vec at 1200: 13.5 / 6.5 / 3.25 / 2.7 M. */
int vbx_pitch_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                  const double *window, double sample_rate, double threshold, double fmin, double fmax,
                  size_t kmax, vbx_pitch *out_cand, int32_t *out_count, int32_t *status);

/* ------------------------------------------------------------------ spectrum.rs: LPC */

/* LPC::lpc(n_coeffs) on autocorrelation rows (Levinson-Durbin, src/spectrum.rs:63-92).
 * r: [F, r_stride] with r_stride >= n_coeffs+1; out: [F, n_coeffs+1] = [1, a1..ap]. */
int vbx_lpc_f64(vbx_ctx *ctx, const double *r, size_t n_frames, size_t r_stride,
                size_t n_coeffs, double *out);
/* LPC::lpc_mut(n_coeffs, ac, kc, tmp) (src/spectrum.rs:62-84): as vbx_lpc_f64, and out_kc: [F, n_coeffs]
 * (optional) receives the reflection coefficients the reference leaves in `kc` (`tmp` is scratch there). */
int vbx_lpc_mut_f64(vbx_ctx *ctx, const double *r, size_t n_frames, size_t r_stride,
                    size_t n_coeffs, double *out_ac, double *out_kc);

/* frame.autocorrelate(n_coeffs+1) [-> .normalize()] -> .lpc(n_coeffs) fused, one pass over the
 * samples (LPCSolver usage, src/spectrum.rs:40-42,470-479).  out_r: [F, n_coeffs+1] (after the
 * optional normalize), out_lpc: [F, n_coeffs+1]; either may be NULL.
 * Ill-conditioned rows (round 6; also the LPC column of vbx_analyze_frames_f64): every row's conditioning is probed -- the
 * recursion repeated on lag sums moved by +-16 eps of r[0] -- and a row such a perturbation moves by more than 1e-6 in the parity
 * metric is recomputed from the frame's samples with the lag sums and the recursion in double-double: the exact row of the f64
 * frame, rounded once, where the reference's own f64 row (src/periodic.rs:284 + src/spectrum.rs:63-84) is 1e-6 .. 3e-4 from it
 * (frame_len <= 4096, n_coeffs <= 31; VBX_LPC_EXACT=0 turns it off). */
int vbx_autocorr_lpc_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len,
                         size_t stride, const double *window, size_t n_coeffs, int normalize,
                         double *out_r, double *out_lpc);

/* LPC::lpc_praat(n_coeffs) per frame (Burg, src/spectrum.rs:94-146).  out: [F, n_coeffs]
 * (no leading 1, sign-flipped as the reference); status[F]: VBX_FRAME_ERR_LPC when denum <= 0.
 * Orders 8, 10, 12, 13, 14, 16 on frames of 256..2048 samples (also inside vbx_find_formants_f64 and vbx_analyze_frames_*): one pass over
 * the frame -- its p + 1 lag sums and first / last p + 1 samples, then an O(p^2) recursion per frame that yields the reference's
 * reflection coefficients (csrc/k_burg_fast.hip).  That recursion is exact in real arithmetic but amplifies the lag sums'
 * rounding by the frame's conditioning, so the kernel bounds its own error per frame: a row is written only if the bound
 * is inside 5e-7 in the parity metric |d| <= 1e-6 max(|a_j|, 1e-6 max|a|); every other frame is computed by the reference's
 * own per-order sums, as all frames of every other order and length are.  HOW MANY frames that is depends on the material:
 * ~1 % of the bench's synthetic 48 kHz signal at order 12; 40 % of a real 44.1 kHz recording at order 13 (68 % without a
 * -70 dB dither: oversampled speech has almost no energy above 8 kHz, its covariance matrix is ill conditioned, and there the
 * one-pass recursion's error is real -- 14 % of such frames are off by more than 1e-7, 5 % by more than 5e-7,
 * tools/experiments/burg_guard_vs_error.py); every pure tone / DC / silent / NaN frame.  The results are the direct
 * recursion's either way; the cost is its speed on those frames (about 4x the one-pass form's instructions per frame;
 * bench.py --signal speech: the pipeline on such a recording runs within 4 % of what the same material would without it).
 * Environment: VBX_BURG_DIRECT=1 (read per call) takes the per-order sums for every frame. */
int vbx_lpc_burg_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len,
                     size_t stride, const double *window, size_t n_coeffs, double *out, int32_t *status);

/* ------------------------------------------------------------------ polynomial.rs */

/* Polynomial::find_roots_mut on F polynomials of `len` coefficients (coefficient of x^j at
 * index j), src/polynomial.rs:92-152: Laguerre from -2-2i with deflation, quadratic/linear
 * tail; roots are written in discovery order, remainder zeroed.  polys: [F, len] in/out. */
int vbx_find_roots_c64(vbx_ctx *ctx, vbx_complex *polys, size_t n_polys, size_t len, int32_t *status);

/* Polynomial::laguerre(start) (src/polynomial.rs:34-72) on F polynomials; out: [F]. */
int vbx_laguerre_c64(vbx_ctx *ctx, const vbx_complex *polys, size_t n_polys, size_t len,
                     vbx_complex start, vbx_complex *out);

/* Polynomial::div_polynomial_mut (src/polynomial.rs:155-195) on F polynomials: polys[f] / (x + others[f]);
 * the quotient is left in polys, rem: [F, len] receives the remainder exactly as the reference leaves it.
 * status: VBX_FRAME_ERR_POLYNOMIAL where others[f] == 0 ("Tried to divide by zero"). */
int vbx_div_polynomial_c64(vbx_ctx *ctx, vbx_complex *polys, const vbx_complex *others, size_t n_polys, size_t len,
                           vbx_complex *rem, int32_t *status);
/* Polynomial::degree / off_low (src/polynomial.rs:26-32) of one HOST polynomial (trivial scans, no device work) */
size_t vbx_degree_c64(const vbx_complex *h_poly, size_t len);
size_t vbx_off_low_c64(const vbx_complex *h_poly, size_t len);

/* The f32 instantiation of Polynomial (Complex<f32>, exercised by the reference's own tests at
 * src/polynomial.rs:336-386): same algorithms in single precision.  polys: [F, len] of {float re, im}. */
typedef struct { float re, im; } vbx_complex32;
int vbx_find_roots_c32(vbx_ctx *ctx, vbx_complex32 *polys, size_t n_polys, size_t len, int32_t *status);
int vbx_laguerre_c32(vbx_ctx *ctx, const vbx_complex32 *polys, size_t n_polys, size_t len,
                     vbx_complex32 start, vbx_complex32 *out);

/* ------------------------------------------------------------------ spectrum.rs: resonances, tracker */

/* ToResonance::to_resonance(sample_rate) per row of roots (src/spectrum.rs:165-210): roots with
 * im >= 0, reflected inside the unit circle, 50 Hz < f < nyquist-50, sorted by frequency.
 * roots: [F, n_roots]; out_res: [F, n_roots] zero padded; out_count[F]. */
int vbx_to_resonance_c64(vbx_ctx *ctx, const vbx_complex *roots, size_t n_rows, size_t n_roots,
                         double sample_rate, vbx_resonance *out_res, int32_t *out_count);

/* EstimateFormants::estimate_formants carried frame to frame = FormantExtractor
 * (src/spectrum.rs:216-369).  The scan is sequential in the reference (the caller passes the
 * previous frame's estimates back in, tests/lib.rs:75-79); here utterances of 384 frames or more are
 * scanned in parallel chunks with exact repair (bit-identical rows, about a millisecond for any batch:
 * utterance length is not a cost).  It is batched per utterance:
 * h_seg_start[n_segments] (HOST array) are the ascending frame indices at which the caller's
 * state is reset to est_init (h_seg_start[0] must be 0; NULL/0 = one segment).  res: [F, n_res]
 * resonance rows exactly as the reference passes them (zero padded); frame_status (optional):
 * frames with status != 0 leave the state untouched.  out: [F, n_est] estimates after each frame. */
int vbx_estimate_formants_f64(vbx_ctx *ctx, const vbx_resonance *res, size_t n_frames, size_t n_res,
                              const int64_t *h_seg_start, size_t n_segments,
                              const vbx_resonance *h_est_init, size_t n_est,
                              const int32_t *frame_status, vbx_resonance *out);

/* vox_box::find_formants(buf, sample_rate, 1.0, .., n_coeffs, .., formants) over F frames
 * (src/lib.rs:40-116): periodic Hanning -> Burg -> reversed complex polynomial -> find_roots_mut
 * -> Resonance::from_root (im > 0) -> sort -> estimate_formants.  `window` must be NULL for
 * rectangular input frames as in tests/lib.rs:71 (the periodic Hanning is applied inside).
 * out_formants: [F, n_est]; out_res (optional): [F, 32] zero padded; out_res_count (optional): [F];
 * out_coeffs (optional): [F, n_coeffs] Burg coefficients; status[F].
 * At the orders 8, 10, 12, 13, 14, 16 the resonance rows come from converged roots of the real polynomial found pair by
 * pair (csrc/k_roots_fast.hip: one Laguerre solve per conjugate pair, deflation by the real quadratic, a Newton step on the
 * original polynomial as polish and check) instead of a replay of find_roots_mut's iteration: the reference runs that
 * iteration to convergence too (20 steps per root) and sorts the result by frequency, so the rows agree to ~1e-10 relative
 * (gate 1e-4; counts and statuses equal).  A frame that fails the check is redone by the reference's own iteration, as
 * every frame of the other orders is.  Environment: VBX_ROOTS_DIRECT=1 (read per call) replays the reference's
 * iteration for every frame; VBX_BURG_DIRECT=1 see vbx_lpc_burg_f64. */
int vbx_find_formants_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len,
                          size_t stride, double sample_rate, size_t n_coeffs,
                          const int64_t *h_seg_start, size_t n_segments,
                          const vbx_resonance *h_est_init, size_t n_est,
                          vbx_resonance *out_formants, vbx_resonance *out_res, int32_t *out_res_count,
                          double *out_coeffs, int32_t *status);

/* ------------------------------------------------------------------ spectrum.rs: MFCC */

/* MFCC::mfcc(num_coeffs, (lo, hi), sample_rate) per frame (src/spectrum.rs:401-441).
 * out: [F, num_coeffs]; status[F]: VBX_FRAME_ERR_PANIC when a mel bin exceeds the spectrum.
 * Which kernel runs depends on the frame length (results within 1e-6 of the reference's arithmetic in every case): a length
 * that is (half of) a transform's takes the fused kernels' forward transform; other lengths from 513 samples take that transform
 * of the zero-padded frame with the frame's DFT bins interpolated from it (vbx_analyze_frames_f64 below explains the
 * interpolation; design error < 1e-14 of the largest bin) where that is the fastest form, the matrix-core two-stage DFT where
 * the length factors suitably and is below 1400 samples, the chirp-z kernel where the filters reach above a quarter of the
 * sampling rate, Goertzel below 600 samples.  VBX_MFCC_INTERP=0 in the environment: no interpolated form anywhere. */
int vbx_mfcc_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                 const double *window, size_t num_coeffs, double lo_hz, double hi_hz,
                 double sample_rate, double *out, int32_t *status);
/* The mel filter bank's bins for that call, on the host: h_bins[num_coeffs + 2] = floor((frame_len + 1) * hz / sample_rate)
 * at num_coeffs + 2 mel-spaced points (src/spectrum.rs:411-414; two points lie beyond hi_hz, and the scale is frame_len + 1:
 * Q14).  Returns 1 (not an error code) when the geometry makes the reference panic on every frame -- a bin beyond the
 * spectrum or descending bins -- which vbx_mfcc_f64 reports as VBX_FRAME_ERR_PANIC per frame. */
int vbx_mfcc_bins(size_t frame_len, size_t num_coeffs, double lo_hz, double hi_hz, double sample_rate, int32_t *h_bins);

/* dct (src/spectrum.rs:384-398) on rows: in/out [F, n]. */
int vbx_dct_f64(vbx_ctx *ctx, const double *in, size_t n_rows, size_t n, double *out);

/* ------------------------------------------------------------------ front end (N2, N3) */

/* 16-bit PCM -> f64 as the reference's tests read WAV data: sample / (i32::MAX >> (32 - bits)) = / 32767
 * (tests/lib.rs:17-19).  pcm: n int16 samples on the device; out: n doubles.  Framing is then the
 * (stride = hop) view of `out`: window::Windower::{rectangle,hanning} without copying frames. */
int vbx_pcm16_to_f64(vbx_ctx *ctx, const int16_t *pcm, size_t n_samples, double *out);

/* RMS::rms per frame (src/waves.rs:10-23).  out: [F]. */
int vbx_rms_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                const double *window, double *out);

/* Filter::preemphasis(factor) per frame (src/waves.rs:82-96): x[i] += 2*pi*factor * x[i+1], backwards.
 * The reference filters in place; frames of a hop-strided view overlap, so the result is written to the
 * dense batch out: [F, frame_len] (out may equal x when stride == frame_len). */
int vbx_preemphasis_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                        double factor, double *out);

/* The resample front end of find_formants (resample_ratio != 1.0, src/lib.rs:42,57-61):
 * sample 0.10's Linear::new(buf[0], buf[1]) + Converter::scale_sample_hz(.., ratio), take(ceil(ratio*len)).
 * That arithmetic lives in the un-vendored `sample` crate and no reference test runs this branch, so this
 * entry point is "parity unpinned" (it is bit-identical to the oracle's restatement of the crate).
 * out: dense [F, vbx_resampled_len(frame_len, ratio)].  find_formants with a ratio is then
 * vbx_find_formants_f64 on that dense batch (frame_len = stride = resampled length). */
size_t vbx_resampled_len(size_t frame_len, double resample_ratio);
int vbx_resample_linear_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                            double resample_ratio, double *out);

/* VecDeque input (`impl Autocorrelate<T> for VecDeque<T>`, src/periodic.rs:291-304 -- the streaming form of the
 * trait): logical sample i of the deque is ring[(head + i) % capacity].  Copies the Windower view over the deque
 * (frame t = logical samples [t*stride, t*stride + frame_len)) into the dense batch out[F, frame_len], which every
 * entry point above accepts with stride = frame_len.  Requires (n_frames-1)*stride + frame_len <= capacity. */
int vbx_ring_frames_f64(vbx_ctx *ctx, const double *ring, size_t capacity, size_t head, size_t n_frames,
                        size_t frame_len, size_t stride, double *out);

/* ------------------------------------------------------------------ Sample = f32 (N4) */

/* The slice traits are generic over the Sample type: `impl<T: Sample> Autocorrelate<T> for [T]`
 * (src/periodic.rs:276-289), `impl<T: Float> LPC<T> for [T]` (src/spectrum.rs:56), `Normalize` (src/waves.rs:60-76),
 * `MFCC<T>` (src/spectrum.rs:401-409).  These are their f32 instantiation: the arguments mean what they mean in the
 * _f64 entry points with float in place of double (frames, windows, outputs).
 * Two forms of each.  The plain names are REFERENCE-FAITHFUL: every fold, product and quotient the generic code performs in
 * T runs in f32, in the reference's order, with no fused multiply-add (the lag sums of src/periodic.rs:284 as sequential
 * f32 folds, one lane per lag; Levinson and Burg as sequential f32 recursions, one lane per frame; Pitched<f32, f32>::pitch
 * with its lag curve built and normalised in f32 and T = f32 roundings of the candidates) -- the results are the bits
 * the crate returns at f32 (tests/test_gpu_f32.py: equal to the f32 restatement, which no reference test pins: parity
 * unpinned).  The *_f32_wide names widen on load, compute in f64 with the f64 kernels and round once: more accurate and as
 * fast as f64, but not those bits.  MFCC exists only in the wide form (rustfft's f32 arithmetic is not in the tree).
 * vbx_window_table_f32: the f64 table rounded to f32 (the sample crate's own f32 window path is not verifiable
 * here: parity unpinned). */
int vbx_window_table_f32(int kind, size_t n, float *h_out);
int vbx_autocorrelate_f32(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                          size_t stride, const float *window, size_t n_lags, float *out);
int vbx_autocorrelate_f32_wide(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                          size_t stride, const float *window, size_t n_lags, float *out);
int vbx_normalize_f32(vbx_ctx *ctx, float *data, size_t n_rows, size_t n);
int vbx_lpc_mut_f32(vbx_ctx *ctx, const float *r, size_t n_frames, size_t r_stride,
                    size_t n_coeffs, float *out_ac, float *out_kc);
int vbx_lpc_mut_f32_wide(vbx_ctx *ctx, const float *r, size_t n_frames, size_t r_stride,
                    size_t n_coeffs, float *out_ac, float *out_kc);
int vbx_autocorr_lpc_f32(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                         size_t stride, const float *window, size_t n_coeffs, int normalize,
                         float *out_r, float *out_lpc);
int vbx_autocorr_lpc_f32_wide(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                         size_t stride, const float *window, size_t n_coeffs, int normalize,
                         float *out_r, float *out_lpc);
int vbx_lpc_burg_f32(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                     size_t stride, const float *window, size_t n_coeffs, float *out, int32_t *status);
int vbx_lpc_burg_f32_wide(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                     size_t stride, const float *window, size_t n_coeffs, float *out, int32_t *status);
int vbx_mfcc_f32(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len, size_t stride,
                 const float *window, size_t num_coeffs, double lo_hz, double hi_hz,
                 double sample_rate, float *out, int32_t *status);
/* Pitched<f32, f32>::pitch (src/periodic.rs:356-358,396-455 at S = T = f32): as vbx_pitch_f64 with float frames, float
 * parameters and Pitch<f32> candidates.  The lag curve and the refinement run in f64 on the widened frame. */
typedef struct { float frequency; float strength; } vbx_pitch32;
int vbx_pitch_f32(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len, size_t stride,
                  const float *window, float sample_rate, float threshold, float fmin, float fmax,
                  size_t kmax, vbx_pitch32 *out_cand, int32_t *out_count, int32_t *status);
int vbx_pitch_f32_wide(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len, size_t stride,
                  const float *window, float sample_rate, float threshold, float fmin, float fmax,
                  size_t kmax, vbx_pitch32 *out_cand, int32_t *out_count, int32_t *status);

/* ------------------------------------------------------------------ the user's frame loop, fused */

/* What a user of the crate writes per frame (examples/pitch_detection.rs:23-30, tests/lib.rs:71-83,
 * examples/formant_extraction/src/main.rs:72-88), as ONE call over F frames of the Windower view:
 *   hanning frame -> pitch::<Hanning>(sr, threshold, _, _, fmin, fmax)[0]        (PitchExtractor output)
 *   hanning frame -> autocorrelate(lpc_order + 1) -> lpc(lpc_order)              (raw, un-normalised autocorrelation)
 *   rectangle frame -> find_formants(.., 1.0, .., formant_order, .., formants)   (state carried per segment)
 *   hanning frame -> mfcc(mfcc_coeffs, (lo, hi), sr)
 * Each part with order / count 0 is skipped.  The library is free to share work between the parts (one pass over
 * the samples, one spectral transform feeding several of them); results obey the same tolerances as the
 * separate entry points.  MFCC shares the pitch path's transform at EVERY frame length from 513 to 4096 samples: where the
 * length divides the transform's (512, 600, 800, 1024, 1200, 2048, 4096) the frame's DFT bins are bins of the transform; at
 * the other lengths (25 ms at 44.1 kHz = 1102 / 1103 samples, ...) each bin is interpolated from 24-40 of the transform's --
 * the frame fills at most half of it, so its spectrum is oversampled twofold and the interpolation's error is a design
 * parameter: < 1e-14 of the largest bin (tests/test_mfcc_interp_table.py), MFCC values within 1e-11 of the chirp-z kernel's exact arithmetic.
 * (Bins above a quarter of the transform -- mfcc_hi_hz beyond ~sample_rate / 4 at a length just below the transform's half --
 * fall back to vbx_mfcc_f64's kernel beside the fused one; VBX_MFCC_INTERP=0 in the environment forces that everywhere.)
 * Output: one record of vbx_record_doubles(params) doubles per frame,
 *   [ pitch.frequency, pitch.strength | formants[n_est] {frequency, bandwidth} | mfcc[mfcc_coeffs] | lpc[lpc_order + 1] ]
 * at out_records + f * record_ld (record_ld even, >= the record size; 16-byte aligned base): the fixed-size
 * per-frame record that the multi-GPU gather below moves.  status3 (optional): [3, F] = pitch / formant / mfcc
 * status rows.  Asynchronous on the context's stream (a second, context-owned stream is used inside and joined). */
typedef struct {
    double sample_rate;
    double pitch_threshold, pitch_fmin, pitch_fmax;
    size_t lpc_order;
    size_t formant_order;
    size_t n_est;
    vbx_resonance est_init[VBX_FORMANT_SLOTS];
    size_t mfcc_coeffs;
    double mfcc_lo_hz, mfcc_hi_hz;
} vbx_analysis_params;
size_t vbx_record_doubles(const vbx_analysis_params *h_params);
int vbx_analyze_frames_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                           const vbx_analysis_params *h_params, const int64_t *h_seg_start, size_t n_segments,
                           double *out_records, size_t record_ld, int32_t *status3);

/* The same frame loop on 16-bit PCM: what a WAV reader hands the reference's callers before the `as f64 / 32767` of
 * tests/lib.rs:17-19 and examples/formant_extraction/src/main.rs.  `pcm` holds the recording's samples (device memory);
 * frame t is pcm[t*stride .. +frame_len), widened in registers exactly as vbx_pcm16_to_f64 would (s / 32767, correctly
 * rounded), so the records are BIT-IDENTICAL to vbx_pcm16_to_f64 followed by vbx_analyze_frames_f64 -- at a quarter of
 * the input bytes (960 instead of 3840 new bytes per 48 kHz / 10 ms frame): the form for host-fed operation, where the
 * samples cross PCIe.  Full 1200-sample frames with lpc_order in {0, 12} read the PCM directly; other shapes are widened
 * into a context-owned f64 copy of the view first. */
int vbx_analyze_frames_pcm16(vbx_ctx *ctx, const int16_t *pcm, size_t n_frames, size_t frame_len, size_t stride,
                             const vbx_analysis_params *h_params, const int64_t *h_seg_start, size_t n_segments,
                             double *out_records, size_t record_ld, int32_t *status3);

/* ------------------------------------------------------------------ multi-GPU: frame-range sharding */

/* The reference has no distribution of any kind; frames are independent (the tracker per utterance), so a long
 * recording shards by contiguous frame ranges, one process per GPU, and the only exchange is ONE gather of the
 * fixed-size per-frame records to a destination rank: grouped ncclSend / ncclRecv (RCCL), each peer's payload
 * crossing its own xGMI link.  No reduction, no all-to-all.
 *
 * vbx_shard_range: frames [*lo, *hi) of rank `rank`: the even split (first ranks take the remainder); with h_seg_start
 * (ascending utterance starts, h_seg_start[0] == 0) a cut moves up to an utterance start that lies within 1/32 of a shard
 * after it.  A cut INSIDE an utterance is fine: ONE long utterance -- what the reference's user loop over a file produces
 * (tests/lib.rs:75-79, src/spectrum.rs:357-369) -- splits evenly, and its formant track is carried across the cut (below).
 * vbx_shard_samples: the samples [*s0, *s1) those frames read, i.e. including the frame_len - hop halo. */
int vbx_shard_range(size_t n_frames, int world, int rank, const int64_t *h_seg_start, size_t n_segments,
                    size_t *lo, size_t *hi);
int vbx_shard_samples(size_t lo, size_t hi, size_t frame_len, size_t hop, size_t *s0, size_t *s1);

/* The formant tracker is the one sequential step of the path (EstimateFormants, src/spectrum.rs:232-333: the estimates after
 * frame t feed frame t + 1).  A rank whose range starts inside an utterance therefore
 *   1. analyses `warm` extra frames before its range, frames [lo - warm, hi), starting the tracker from the initial estimates:
 *      the tracker forgets -- after a few dozen frames its state no longer depends on where it started -- so the rows of
 *      [lo, hi) are, almost always, already the sequential scan's;
 *   2. receives the formant row its predecessor ENDS with (the true state before frame lo), compares it bit for bit with its
 *      own row of frame lo - 1, and where they differ redoes the scan from the true state until it meets rows it already
 *      holds (vbx_track_stitch_f64; over a communicator: vbx_comm_stitch_tracks_f64, which also passes the rank's own last
 *      row on).  The result is the single-process scan, bit for bit, for every world size.
 * vbx_shard_plan: lo, hi as vbx_shard_range; warm = min(frames since the utterance's start, VBX_SHARD_WARM_FRAMES);
 * stop = index, counted from frame lo - warm, at which the utterance that holds frame lo ends inside the shard (or the
 * shard's end); continues_prev: the utterance starts more than `warm` frames before lo (the state must come from rank - 1);
 * continues_next: the same for the next rank's first frame.
 * vbx_shard_local_segments: the utterance starts of frames [lo - warm, hi), re-based to the shard (first entry 0): the
 * h_seg_start of the rank's vbx_analyze_frames_f64 / vbx_find_formants_f64 call.  *n_out = entries needed (h_out may be NULL). */
#define VBX_SHARD_WARM_FRAMES 64
typedef struct {
    size_t lo, hi, warm, stop;
    int continues_prev, continues_next;
} vbx_shard_plan_t;
int vbx_shard_plan(size_t n_frames, int world, int rank, const int64_t *h_seg_start, size_t n_segments, vbx_shard_plan_t *h_out);
int vbx_shard_local_segments(const vbx_shard_plan_t *h_plan, const int64_t *h_seg_start, size_t n_segments,
                             int64_t *h_out, size_t cap, size_t *n_out);
/* Step 2 on one device: `formants` are the rows the LAST vbx_find_formants_f64 (out_formants, formants_ld = 2 n_est) or
 * vbx_analyze_frames_* call (out_records + 2, formants_ld = record_ld) on this context wrote, n_frames of them (call it right
 * after that call: it reads the resonance rows the context still holds); d_state_in (device, n_est entries) is the true state
 * before frame `first`; rows [first, stop) are corrected where needed.  *d_changed (device, optional) = rows rewritten. */
int vbx_track_stitch_f64(vbx_ctx *ctx, vbx_resonance *formants, size_t n_frames, size_t formants_ld, size_t first, size_t stop,
                         const vbx_resonance *d_state_in, int32_t *d_changed);

typedef struct vbx_comm vbx_comm;
#define VBX_UNIQUE_ID_BYTES 128
#define VBX_COMM_SLOTS 4
/* ncclGetUniqueId on the root; the caller ships the 128 bytes to the other ranks (MPI, TCP store, file ...). */
int vbx_comm_unique_id(void *h_id);
/* One communicator per (context, process): rank `rank` of `world` on the context's device.  Collective call. */
int vbx_comm_create(vbx_ctx *ctx, const void *h_id, int world, int rank, vbx_comm **out);
void vbx_comm_destroy(vbx_comm *comm);
/* Gathers per-frame records to rank dst: rank r contributes h_rows[r] rows of row_doubles doubles (`local`, device),
 * which land in `out` (device, on dst only) at row offset h_rows[0] + .. + h_rows[r-1].  On dst, `local` may point
 * into `out` at its own offset (kernels write their records in place: no copy).  The transfer is queued on the
 * communicator's own stream behind the work already queued on the context's stream, so the context's next
 * batch overlaps it; `slot` in [0, VBX_COMM_SLOTS) names the buffer being sent for vbx_comm_wait. */
int vbx_gather_records_f64(vbx_ctx *ctx, vbx_comm *comm, const double *local, const int64_t *h_rows,
                           size_t row_doubles, int dst, double *out, int slot);
/* Step 2 across ranks (RCCL, on the communicator's stream, behind the work queued on the context's stream): receives the
 * previous rank's last formant row when h_plan->continues_prev (it sends it after its own stitch: the ranks of one utterance
 * form a chain of 2 n_est doubles each over the direct xGMI links), corrects this rank's rows, and sends this rank's last row
 * on when h_plan->continues_next.  n_frames = hi - lo + warm rows as in vbx_track_stitch_f64.  Queue the record gather after
 * it with the same `slot`: vbx_comm_wait(slot) then covers both.  A rank that receives makes the context's stream wait (on
 * the device) until its stitch is through: the repair reads the resonance rows the context holds, which the context's next
 * call overwrites.  Errors: argument / plan checks run before the first RCCL call; once this rank's receive is posted its send
 * is posted too whatever fails in between (the next rank waits for it), and the first error is returned afterwards -- a
 * VBX_E_* from this call means the chain's rows are not to be trusted: destroy the communicator on EVERY rank. */
int vbx_comm_stitch_tracks_f64(vbx_ctx *ctx, vbx_comm *comm, vbx_resonance *formants, size_t n_frames, size_t formants_ld,
                               const vbx_shard_plan_t *h_plan, int32_t *d_changed, int slot);
/* The transfer list of that gather as one rank sees it, on the host (no GPU, no RCCL: what vbx_gather_records_f64 posts,
 * exposed so that a caller -- and the CPU tests -- can check the layout for any world size and uneven h_rows):
 * for every rank r, h_offset[r] = element offset (doubles) of rank r's rows in `out`, h_count[r] = doubles rank r
 * contributes; h_op[r] = what THIS rank does for peer r: VBX_GATHER_NONE, VBX_GATHER_RECV (this rank is dst and r sends),
 * VBX_GATHER_SEND (r == dst and this rank has rows), VBX_GATHER_COPY (r == rank == dst: device copy unless in place).
 * Arrays of `world` entries; any of the three may be NULL. */
enum { VBX_GATHER_NONE = 0, VBX_GATHER_RECV = 1, VBX_GATHER_SEND = 2, VBX_GATHER_COPY = 3 };
int vbx_gather_plan(const int64_t *h_rows, int world, int rank, int dst, size_t row_doubles,
                    int64_t *h_offset, int64_t *h_count, int32_t *h_op);
/* Live communicators of this process (the bench prints it: exactly one RCCL instance per rank). */
int vbx_comm_live_count(void);
/* Makes the context's stream wait (on the device, not the host) until the gather that used `slot` has finished:
 * call before overwriting that buffer. */
int vbx_comm_wait(vbx_ctx *ctx, vbx_comm *comm, int slot);
/* Host waits for every queued gather. */
int vbx_comm_sync(vbx_comm *comm);
/* Loopback self-test: a grouped ncclSend/ncclRecv of n doubles from this rank to itself on the communicator's
 * stream, verified on the host.  Exercises the RCCL path on a single GPU.  Returns 0 when the data arrived intact. */
int vbx_comm_selftest(vbx_ctx *ctx, vbx_comm *comm, size_t n_doubles);

/* ------------------------------------------------------------------ bench utility */

/* Deterministic speech-like synthetic audio (DESIGN.md "synthetic signal"): samples
 * [sample_offset, sample_offset + n_samples) of an endless 48 kHz-style stream defined in
 * closed form per sample, so any shard can be generated in place on its own GPU. */
int vbx_synth_speech_f64(vbx_ctx *ctx, double *out, size_t n_samples, uint64_t sample_offset,
                         double sample_rate, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* VOXBOX_HIP_H */
