/*
 * vbx_oracle.h -- CPU restatement of the vox_box 0.3.0 per-frame DSP path (f64).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the timed CPU baseline.
 * The shipped path (vox_box.rs_amd/) never links, loads or calls it.
 *
 * Every function cites the reference file:line (under /root/reference) whose
 * behaviour it restates, INCLUDING the quirks of SURVEY.md Appendix A.
 *
 * Parity pinning: the reference is Rust and cannot be built in this image (no
 * rustc/cargo, six un-vendored crates).  The oracle is pinned against every
 * known-answer test the reference's own test-suite holds for this path
 * (tests/test_oracle_kat.py lists them with file:line).  Routines the
 * reference does not pin numerically (MFCC values, find_formants end-to-end,
 * sinc/Brent beyond the 150 Hz case) are "restatement-derived" and say so.
 *
 * Third-party arithmetic restated from the published crate sources (not in
 * /root/reference): sample 0.10 (Window phase accumulation, Hanning::at_phase,
 * Windower), num-complex 0.2 (mul/div/norm/sqrt/to_polar/inv), rustfft 1.0
 * (unnormalised forward DFT; restated as the mathematical DFT).
 */
#ifndef VBX_ORACLE_H
#define VBX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } vbxo_c64;
typedef struct { double frequency, strength; } vbxo_pitch_t;      /* periodic.rs:306-310 */
typedef struct { double frequency, bandwidth; } vbxo_resonance_t; /* spectrum.rs:149-154 */

enum {
    VBXO_OK = 0,
    VBXO_ERR_LPC = 1,         /* VoxBoxError::LPC, spectrum.rs:123-125 */
    VBXO_ERR_POLYNOMIAL = 2,  /* VoxBoxError::Polynomial, polynomial.rs:95,123 */
    VBXO_ERR_NAN = 3,         /* partial_cmp().unwrap() panic, periodic.rs:453 */
    VBXO_ERR_PANIC = 4,       /* any other panic of the reference (OOB index, assert) */
    VBXO_ERR_WORKSPACE = 5    /* VoxBoxError::Workspace, lib.rs:46-48 */
};

#define VBXO_MAX_RESONANCES 32   /* lib.rs:26 */
#define VBXO_FORMANT_SLOTS 6     /* spectrum.rs:228 */

/* ---- window tables (sample 0.10 semantics) ---- */
void vbxo_window_hanning(double *w, size_t n);          /* Window::<Hanning>::new(n), accumulated phase */
void vbxo_window_hanning_lag(double *w, size_t n);      /* periodic.rs:236-248 via :400 */
void vbxo_window_hanning_periodic(double *w, size_t n); /* lib.rs:66-70 */
void vbxo_sine(double *x, size_t n, double rate, double hz); /* signal::rate(rate).const_hz(hz).sine() */

/* ---- waves.rs ---- */
double vbxo_max_amplitude(const double *x, size_t n);   /* waves.rs:25-58 */
void vbxo_normalize(double *x, size_t n);               /* waves.rs:60-76 */
double vbxo_rms(const double *x, size_t n);             /* waves.rs:10-23 */
void vbxo_preemphasis(double *x, size_t n, double factor); /* waves.rs:82-96 */

/* ---- periodic.rs ---- */
void vbxo_autocorrelate(const double *x, size_t n, double *coeffs, size_t n_lags); /* :276-289 */
int vbxo_interpolate_sinc(const double *y, size_t ylen, long offset, size_t nx,
                          double x, size_t max_depth, double *out);               /* :29-87 */
int vbxo_improve_extremum_sinc(const double *y, size_t ylen, long offset, size_t nx,
                               double ixmid, size_t depth, double *xmid, double *ymid); /* :192-229 */
/* Pitched::pitch (:396-455).  Writes min(count, cap) candidates (sorted, stable,
 * descending strength), returns status, *count = full candidate count. */
int vbxo_pitch(const double *x, size_t n, double sample_rate, double threshold,
               double fmin, double fmax, vbxo_pitch_t *out, size_t cap, size_t *count);

/* ---- spectrum.rs: LPC ---- */
void vbxo_lpc(const double *r, size_t n_coeffs, double *ac /* n_coeffs+1 */);      /* :63-92 */
int vbxo_lpc_burg(const double *x, size_t n, size_t n_coeffs, double *coeffs);     /* :94-146 */

/* ---- polynomial.rs ---- */
size_t vbxo_degree(const vbxo_c64 *p, size_t len);                                 /* :26-28 */
size_t vbxo_off_low(const vbxo_c64 *p, size_t len);                                /* :30-32 */
vbxo_c64 vbxo_laguerre(const vbxo_c64 *p, size_t len, vbxo_c64 start);             /* :34-72 */
int vbxo_find_roots_mut(vbxo_c64 *p, size_t len);                                  /* :92-152 */
int vbxo_div_polynomial_mut(vbxo_c64 *p, size_t len, vbxo_c64 other, vbxo_c64 *rem); /* :155-195 */
/* find_roots (:79-89): copies, solves, pops trailing zeros. returns status; *n_roots out */
int vbxo_find_roots(const vbxo_c64 *p, size_t len, vbxo_c64 *roots, size_t *n_roots);

/* ---- polynomial.rs, Complex<f32> instantiation (vbx_oracle_f32.c; reference tests :336-386) ---- */
typedef struct { float re, im; } vbxo_c32;
vbxo_c32 vbxo_laguerre_f32(const vbxo_c32 *p, size_t len, vbxo_c32 start);
int vbxo_find_roots_mut_f32(vbxo_c32 *p, size_t len);
int vbxo_find_roots_f32(const vbxo_c32 *p, size_t len, vbxo_c32 *roots, size_t *n_roots);

/* ---- spectrum.rs: resonances, tracker ---- */
int vbxo_resonance_from_root(vbxo_c64 root, double sample_rate, vbxo_resonance_t *out); /* :165-193, 1 = Some */
size_t vbxo_to_resonance(const vbxo_c64 *roots, size_t n, double sample_rate, vbxo_resonance_t *out); /* :199-210 */
void vbxo_estimate_formants(vbxo_resonance_t *est, size_t n_est,
                            const vbxo_resonance_t *res, size_t n_res);            /* :232-333 */

/* ---- lib.rs ---- */
/* find_formants with resample_ratio == 1.0 (:40-116).  x is NOT modified (the
 * reference copies into resampled_buf, :63).  res_out (32 entries, zero padded,
 * sorted as :105-110) and coeffs_out (n_coeffs, Burg) are optional (NULL ok). */
int vbxo_find_formants(const double *x, size_t n, double sample_rate, size_t n_coeffs,
                       vbxo_resonance_t *formants, size_t n_formants,
                       vbxo_resonance_t *res_out, double *coeffs_out);

/* lib.rs:57-61: the resample front end of find_formants (resample_ratio != 1.0):
 *   Linear::new(buf[0], buf[1]) + Converter::scale_sample_hz(rest, linear, ratio), take(ceil(ratio*len)).
 * sample 0.10 arithmetic restated from the crate (NOT in /root/reference, and no reference test runs this
 * branch: PARITY UNPINNED): interpolation_value accumulates 1/ratio, whole steps advance (left,right),
 * out = (right-left)*value + left, equilibrium (0) once the source is exhausted.
 * out must hold vbxo_resampled_len(n, ratio) samples. */
size_t vbxo_resampled_len(size_t n, double ratio);
void vbxo_resample_linear(const double *x, size_t n, double ratio, double *out);
/* find_formants with any ratio (ratio == 1.0 -> vbxo_find_formants) */
int vbxo_find_formants_ratio(const double *x, size_t n, double sample_rate, double ratio, size_t n_coeffs,
                             vbxo_resonance_t *formants, size_t n_formants);

/* ---- spectrum.rs: MFCC ---- */
double vbxo_hz_to_mel(double hz);                                                  /* :375-377 */
double vbxo_mel_to_hz(double mel);                                                 /* :379-381 */
void vbxo_dct(const double *signal, size_t n, double *coeffs);                     /* :391-398 */
void vbxo_mfcc_bins(size_t n, size_t num_coeffs, double lo, double hi, double sr, size_t *bins /* num_coeffs+2 */);
/* use_fft: 0 = direct DFT of the needed bins (most accurate), 1 = mixed-radix FFT (timing baseline) */
int vbxo_mfcc(const double *x, size_t n, size_t num_coeffs, double lo, double hi,
              double sample_rate, double *out, int use_fft);                       /* :410-440 */
void vbxo_fft(const vbxo_c64 *in, vbxo_c64 *out, size_t n);                        /* rustfft FFT::new(n,false).process */

/* ---- instrumentation for the flop model (bench/DESIGN only) ---- */
typedef struct {
    uint64_t autocorr_macs, sinc_terms, sinc_evals, brent_calls, candidates;
} vbxo_counters_t;
void vbxo_counters_reset(void);
void vbxo_counters_get(vbxo_counters_t *out);

#ifdef __cplusplus
}
#endif
/* improve_extremum with every Interpolation arm (0 None, 1 Parabolic, 2 Sinc(depth)) and is_max, src/periodic.rs:192-229 */
int vbxo_improve_extremum(const double *y, size_t ylen, long offset, size_t nx, double ixmid, int interp, size_t depth,
                          int is_max, double *xmid, double *ymid);

/* Sample = f32 instantiation of the slice traits (vbx_oracle_f32.c; parity unpinned: no reference test runs them) */
void vbxo_autocorrelate_f32(const float *x, size_t n, float *coeffs, size_t n_lags);
void vbxo_normalize_f32(float *x, size_t n);
void vbxo_lpc_f32(const float *r, size_t n_coeffs, float *ac /* n_coeffs+1 */, float *kc /* n_coeffs or NULL */);
int vbxo_lpc_burg_f32(const float *x, size_t n, size_t n_coeffs, float *coeffs);
int vbxo_mfcc_f32(const float *x, size_t n, size_t num_coeffs, double lo, double hi, double sample_rate, float *out);
int vbxo_pitch_f32(const float *x, size_t n, float sample_rate, float threshold, float fmin, float fmax,
                   vbxo_pitch_t *out, size_t cap, size_t *count);

#endif
