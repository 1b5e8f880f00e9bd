/*
 * vbx_oracle.c -- CPU restatement (plain C, f64) of the vox_box 0.3.0 hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see vbx_oracle.h.  Never linked into the product.
 * Each routine follows the cited reference lines literally, quirks included
 * (SURVEY.md Appendix A, Q1..Q14).  Summation ORDER follows the reference too,
 * so this file is also the "reference CPU path" that bench.py times.
 */
#include "vbx_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846264338327950288
#endif

static __thread vbxo_counters_t g_cnt;
void vbxo_counters_reset(void) { memset(&g_cnt, 0, sizeof g_cnt); }
void vbxo_counters_get(vbxo_counters_t *out) { *out = g_cnt; }

/* ------------------------------------------------------------------------ */
/* sample 0.10: signal::Phase / window::Window / Sine (not in /root/reference) */
/* ------------------------------------------------------------------------ */

/* Phase::next_phase(): yields the current phase, then next = (next + step) % 1.0.
 * Window::new(len) uses rate(len - 1).const_hz(1.0) -> step = 1.0 / (len - 1). */
static void phase_table(double *phi, size_t n, double step) {
    double next = 0.0;
    for (size_t i = 0; i < n; i++) {
        phi[i] = next;
        next = fmod(next + step, 1.0);
    }
}

/* window::Hanning::at_phase = 0.5 * (1 - cos(2*pi*phase)); Window::<Hanning>::new(n)
 * as driven by Windower::hanning (examples/pitch_detection.rs:23, periodic.rs:493). */
void vbxo_window_hanning(double *w, size_t n) {
    phase_table(w, n, 1.0 / ((double)n - 1.0));
    for (size_t i = 0; i < n; i++) w[i] = 0.5 * (1.0 - cos(w[i] * (M_PI * 2.0)));
}

/* HanningLag::at_phase, periodic.rs:239-247, driven by Window::new(N).take(N) at :400 (Q3). */
void vbxo_window_hanning_lag(double *w, size_t n) {
    phase_table(w, n, 1.0 / ((double)n - 1.0));
    const double pi_2 = M_PI * 2.0;
    for (size_t i = 0; i < n; i++) {
        double phase = w[i];
        double v = phase * pi_2;
        w[i] = (1.0 - phase) * (2.0 / 3.0 + (1.0 / 3.0) * cos(v)) + (1.0 / pi_2) * sin(v);
    }
}

/* lib.rs:66-70: Hanning::at_phase(idx * (1/len)) -- "periodic" Hanning. */
void vbxo_window_hanning_periodic(double *w, size_t n) {
    double len_inv = 1.0 / (double)n;
    for (size_t i = 0; i < n; i++) {
        double phase = (double)i * len_inv;
        w[i] = 0.5 * (1.0 - cos(phase * (M_PI * 2.0)));
    }
}

/* signal::rate(rate).const_hz(hz).sine(): sample_i = sin(2*pi*phase_i), phase accumulated. */
void vbxo_sine(double *x, size_t n, double rate, double hz) {
    phase_table(x, n, hz / rate);
    for (size_t i = 0; i < n; i++) x[i] = sin((M_PI * 2.0) * x[i]);
}

/* ------------------------------------------------------------------------ */
/* waves.rs                                                                  */
/* ------------------------------------------------------------------------ */

/* waves.rs:29-37 */
static double amplitude(double s) { return (s < 0.0) ? s * -1.0 : s; }

/* waves.rs:44-58: fold keeping acc unless amp is strictly Greater (NaN keeps acc). */
double vbxo_max_amplitude(const double *x, size_t n) {
    double acc = amplitude(x[0]);
    for (size_t i = 1; i < n; i++) {
        double amp = amplitude(x[i]);
        if (amp > acc) acc = amp;
    }
    return acc;
}

/* waves.rs:68-75 (Q2: max over ALL entries). */
void vbxo_normalize(double *x, size_t n) {
    double scale = 1.0 / vbxo_max_amplitude(x, n);
    for (size_t i = 0; i < n; i++) x[i] = x[i] * scale;
}

/* waves.rs:14-22 */
double vbxo_rms(const double *x, size_t n) {
    double sum = 0.0;
    for (size_t i = 0; i < n; i++) sum = sum + x[i] * x[i];
    return sqrt(sum / (double)n);
}

/* waves.rs:86-95: backwards recursion x[i] += (2*pi*factor) * x[i+1] (already updated). */
void vbxo_preemphasis(double *x, size_t n, double factor) {
    double last = x[n - 1];
    double filter = 2.0 * M_PI * factor;
    for (size_t k = n - 1; k-- > 0;) {
        x[k] = x[k] + last * filter;
        last = x[k];
    }
}

/* ------------------------------------------------------------------------ */
/* periodic.rs                                                               */
/* ------------------------------------------------------------------------ */

/* periodic.rs:279-288 (Q1): fold seeded with self[0]; terms i = 1 .. n-lag-1. */
void vbxo_autocorrelate(const double *x, size_t n, double *coeffs, size_t n_lags) {
    for (size_t lag = 0; lag < n_lags; lag++) {
        double accum = x[0];
        /* enumerate().take(len - lag).skip(1): i in 1 .. len-lag */
        for (size_t i = 1; i + lag < n; i++) accum = accum + x[i] * x[i + lag];
        coeffs[lag] = accum;
        g_cnt.autocorr_macs += (n > lag + 1) ? (n - lag - 1) : 0;
    }
}

/* periodic.rs:29-87 (Q6, Q7).  Index arithmetic on `offset as usize + ...` wraps in
 * release Rust; restated with signed arithmetic and an explicit bounds check that
 * maps the reference's out-of-bounds panic to VBXO_ERR_PANIC. */
int vbxo_interpolate_sinc(const double *y, size_t ylen, long offset, size_t nx,
                          double x, size_t max_depth, double *out) {
    /* `x.floor() as usize` saturates at 0 for negative x (Rust >= 1.45) */
    double fl = floor(x);
    size_t nl = (fl > 0.0) ? (size_t)fl : 0; /* NaN -> 0 */
    size_t nr = nl + 1;
    double phil = x - (double)nl;
    double phir = 1.0 - phil;
    double result = 0.0;
    long idx;

    g_cnt.sinc_evals++;
    if (nx < 1) { *out = NAN; return VBXO_OK; }                                 /* :38 */
    if (x > (double)nx) {                                                       /* :39 */
        idx = offset + (long)nx - 1;
        if (idx < 0 || (size_t)idx >= ylen) return VBXO_ERR_PANIC;
        *out = y[idx]; return VBXO_OK;
    }
    if (x < 0.0) { *out = y[0]; return VBXO_OK; }                               /* :40 */
    if (fabs(x - (double)nl) < 1.0e-10) {                                       /* :41 */
        idx = offset + (long)nl;
        if (idx < 0 || (size_t)idx >= ylen) return VBXO_ERR_PANIC;
        *out = y[idx]; return VBXO_OK;
    }
    if (fabs(x - (double)nr) < 1.0e-10) {                                       /* :42 */
        idx = offset + (long)nr;
        if (idx < 0 || (size_t)idx >= ylen) return VBXO_ERR_PANIC;
        *out = y[idx]; return VBXO_OK;
    }

    /* :46-52 clip max_depth to offset + nr at the lowest point */
    if ((offset + (long)nr) < (long)max_depth) {
        if ((offset + (long)nr) < 0) max_depth = 0;
        else max_depth = (size_t)(offset + (long)nr);
    }
    /* :55-57 second clip (only reachable for tiny frames) */
    if ((offset + (long)nl + (long)max_depth) >= (long)nx) {
        long d = (long)nx - offset + (long)nl - 1;
        max_depth = (size_t)d;
    }

    for (size_t n = 0; n < max_depth + 1; n++) {                                /* :59 */
        {   /* :61-72 "left" */
            double a = M_PI * (phil + (double)n);
            long lag_val = (long)(int)offset + (long)(int)nr - (long)(int)n;
            if (lag_val < 0) lag_val = 0;
            if ((size_t)lag_val >= ylen) return VBXO_ERR_PANIC;
            double r_lag = y[lag_val];
            double first = sin(a) / a;
            double second = 0.5 + 0.5 * cos(a / (phil + (double)max_depth));
            result += r_lag * first * second;
        }
        {   /* :74-83 "right" */
            double a = M_PI * (phir + (double)n);
            long lag_val = (long)(int)offset + (long)(int)nl + (long)(int)n;
            if (lag_val < 0) lag_val = 0;
            if (lag_val >= (long)ylen) lag_val = (long)ylen - 1;
            double r_lag = y[lag_val];
            double first = sin(a) / a;
            double second = 0.5 + 0.5 * cos(a / (phir + (double)max_depth));
            result += r_lag * first * second;
        }
    }
    g_cnt.sinc_terms += 2 * (max_depth + 1);
    *out = result;
    return VBXO_OK;
}

typedef struct {
    const double *y; size_t ylen; long offset; size_t depth; size_t ixmax; int status; int negate;
} brent_params;

/* closure at periodic.rs:216-223: the interpolant for is_max == true (NOT negated, Q8), its negative otherwise. */
static double brent_f(double x, brent_params *p) {
    double out = NAN;
    int st = vbxo_interpolate_sinc(p->y, p->ylen, p->offset, p->ixmax, x, p->depth, &out);
    if (st != VBXO_OK) p->status = st;
    return p->negate ? -out : out;
}

/* periodic.rs:103-188: Brent golden/parabolic MINIMISER, tol 1e-10, <= 60 iterations. */
static double brent_maximize(double a, double b, brent_params *params, double tol, double *fx) {
    const double golden = 1. - 0.6180339887498948482045868343656381177203091798057628621;
    const double sqrt_epsilon = sqrt(DBL_EPSILON);
    const int itermax = 60;

    double v = a + golden * (b - a);
    double fv = brent_f(v, params);
    double x = v;
    double w = v;
    *fx = fv;
    double fw = fv;
    g_cnt.brent_calls++;

    for (int it = 1; it < itermax + 1; it++) {
        double range = b - a;
        double middle_range = (a + b) * 0.5;
        double tol_act = sqrt_epsilon * fabs(x) + tol / 3.;

        if (fabs(x - middle_range) + range * 0.5 <= 2. * tol_act) return x;

        double new_step = (x < middle_range) ? golden * (b - x) : golden * (a - x);

        if (fabs(x - w) >= tol_act) {
            double t = (x - w) * (*fx - fv);
            double q = (x - v) * (*fx - fw);
            double p = (x - v) * q - (x - w) * t;
            q = 2. * q - t;
            if (q > 0.) p = -p; else q = -q;
            if (fabs(p) < fabs(new_step * q) &&
                p > q * (a - x + 2. * tol_act) &&
                p < q * (b - x - 2. * tol_act)) {
                new_step = p / q;
            }
        }

        if (fabs(new_step) < tol_act) new_step = (new_step > 0.) ? tol_act : -tol_act;

        {
            double t = x + new_step;
            double ft = brent_f(t, params);
            if (ft <= *fx) {
                if (t < x) b = x; else a = x;
                v = w; w = x; x = t;
                fv = fw; fw = *fx; *fx = ft;
            } else {
                if (t < x) a = t; else b = t;
                if (ft <= fw || fabs(w - x) < DBL_EPSILON) {
                    v = w; w = t;
                    fv = fw; fw = ft;
                } else if (ft <= fv || fabs(v - x) < DBL_EPSILON || fabs(v - w) < DBL_EPSILON) {
                    v = t;
                    fv = ft;
                }
            }
        }
    }
    return x;
}

/* periodic.rs:192-229, Interpolation::Sinc(depth) arm, is_max = true. */
int vbxo_improve_extremum_sinc(const double *y, size_t ylen, long offset, size_t nx,
                               double ixmid, size_t depth, double *xmid, double *ymid) {
    if (ixmid == 0.) { *xmid = 0.; *ymid = y[0]; return VBXO_OK; }                 /* :193 */
    if (ixmid >= (double)nx) {                                                      /* :194 */
        if (nx < 1 || nx - 1 >= ylen) return VBXO_ERR_PANIC;
        *xmid = (double)nx; *ymid = y[nx - 1]; return VBXO_OK;
    }
    brent_params p = { y, ylen, offset, depth, nx, VBXO_OK, 0 };
    double a = ixmid - 1., b = ixmid + 1.;
    if (!(a < b)) return VBXO_ERR_PANIC;                                            /* assert, :113 */
    double result = 0.;
    *xmid = brent_maximize(a, b, &p, 1e-10, &result);
    *ymid = result;
    return p.status;
}

/* periodic.rs:192-229 with every arm: interp 0 = Interpolation::None, 1 = Parabolic, 2 = Sinc(depth); is_max as given.
 * Only the Sinc arm with is_max == true is on the pitch path; the others are public API. */
int vbxo_improve_extremum(const double *y, size_t ylen, long offset, size_t nx, double ixmid, int interp, size_t depth,
                          int is_max, double *xmid, double *ymid) {
    if (ylen < 1) return VBXO_ERR_PANIC;
    if (ixmid == 0.) { *xmid = 0.; *ymid = y[0]; return VBXO_OK; }                 /* :193 */
    if (ixmid >= (double)nx) {                                                      /* :194 */
        if (nx < 1 || nx - 1 >= ylen) return VBXO_ERR_PANIC;
        *xmid = (double)nx; *ymid = y[nx - 1]; return VBXO_OK;
    }
    if (interp == 0) { *xmid = 0.; *ymid = y[0]; return VBXO_OK; }                 /* :197-199 */
    if (interp == 1) {                                                              /* :200-207 */
        double fl = floor(ixmid);
        if (!(fl >= 1.) || !(fl + 1. < (double)ylen)) return VBXO_ERR_PANIC;       /* usize underflow / out of bounds (NaN too) */
        size_t i = (size_t)fl;
        double diff = y[i + 1] - y[i - 1];
        double mid = y[i];
        double dy = 0.5 * diff;
        double d2y = 2.0 * mid - diff;
        *xmid = ixmid + dy / d2y;
        *ymid = mid + 0.5 * dy * dy / d2y;
        return VBXO_OK;
    }
    brent_params p = { y, ylen, offset, depth, nx, VBXO_OK, is_max ? 0 : 1 };
    double a = ixmid - 1., b = ixmid + 1.;
    if (!(a < b)) return VBXO_ERR_PANIC;
    double result = 0.;
    *xmid = brent_maximize(a, b, &p, 1e-10, &result);
    *ymid = result;
    return p.status;
}

/* periodic.rs:413-455 (Q4..Q10): everything after the lag curve.  self_lag: 2n entries, [n, 2n) zero (:411).
 * f32 != 0: the Sample / Float types are f32 -- the curve's entries are f32 values, and the frequency arithmetic of
 * :421-432, :443 and :447 rounds to f32 where the generic code computes in T (the sinc / Brent routines are f64 in
 * either instantiation). */
static int pitch_from_lag(const double *self_lag, size_t n, double sample_rate, double threshold,
                          double fmin, double fmax, vbxo_pitch_t *out, size_t cap, size_t *count, int f32) {
    int status = VBXO_OK;
    size_t max_maxima = n / 2 + 2;
    vbxo_pitch_t *maxima = (vbxo_pitch_t *)malloc(max_maxima * sizeof(vbxo_pitch_t));
    size_t n_max = 0;
#define RT(v) (f32 ? (double)(float)(v) : (double)(v))      /* a value of type T */
    const double interpolation_depth = 0.5;
    size_t brent_ixmax = (size_t)floor(interpolation_depth * (double)n);           /* :414 */
    long offset = -(long)brent_ixmax - 1;                                          /* :429,:441 */
    size_t nx = (size_t)((long)brent_ixmax - offset);                              /* :430,:442 */
    size_t ylen = 2 * n;

    /* :417 local_maxima over self_lag[0..brent_ixmax]: windows(3), strict (Q4) */
    for (size_t k = 1; k + 1 < brent_ixmax; k++) {
        if (!(self_lag[k - 1] < self_lag[k] && self_lag[k + 1] < self_lag[k])) continue;
        double peak = self_lag[k], peak_rev = self_lag[k - 1], peak_fwd = self_lag[k + 1];
        double dr = 0.5 * RT(peak_fwd - peak_rev);                                 /* :423 */
        double d2r = 2. * peak - RT(peak_rev - peak_fwd);                          /* :424 (Q5) */
        double freq = RT(sample_rate / RT((double)k + dr / d2r));                  /* :425 */
        double nn = RT(RT(sample_rate / freq) - (double)offset);                   /* :432 */
        double strn = NAN;
        int st = vbxo_interpolate_sinc(self_lag, ylen, offset, nx, nn, 30, &strn); /* :433 */
        if (st != VBXO_OK) { status = st; break; }
        if (strn > 1.) strn = 1. / strn;                                           /* :435 */
        /* :439 filter */
        if (!((freq == 0.0) || (freq > fmin && freq < fmax))) continue;
        g_cnt.candidates++;
        /* :440-450 refine */
        double n2 = RT(RT(sample_rate / freq) - (double)offset);                   /* :443 */
        double xmid = 0., ymid = 0.;
        st = vbxo_improve_extremum_sinc(self_lag, ylen, offset, nx, n2, 1200, &xmid, &ymid); /* :444 */
        if (st != VBXO_OK) { status = st; break; }
        xmid += (double)offset;                                                    /* :445 */
        if (ymid > 1.) ymid = 1. / ymid;                                           /* :446 */
        maxima[n_max].frequency = RT(sample_rate / RT(xmid));                      /* :447 */
        maxima[n_max].strength = RT(ymid);                                         /* :448 */
        n_max++;
    }
#undef RT
    if (status == VBXO_OK) {
        maxima[n_max].frequency = 0.; maxima[n_max].strength = threshold;          /* :452 */
        n_max++;
        /* :453 stable sort, descending strength; partial_cmp().unwrap() panics on NaN (Q10).
         * A 1-element sort performs no comparison. */
        if (n_max > 1)
            for (size_t i = 0; i < n_max; i++)
                if (isnan(maxima[i].strength)) status = VBXO_ERR_NAN;
        if (status == VBXO_OK) {
            for (size_t i = 1; i < n_max; i++) {       /* stable insertion sort */
                vbxo_pitch_t key = maxima[i];
                size_t j = i;
                while (j > 0 && maxima[j - 1].strength < key.strength) { maxima[j] = maxima[j - 1]; j--; }
                maxima[j] = key;
            }
        }
    }
    if (status == VBXO_OK) {
        *count = n_max;
        for (size_t i = 0; i < n_max && i < cap; i++) out[i] = maxima[i];
    } else {
        *count = 0;
    }
    free(maxima);
    return status;
}

/* periodic.rs:396-455 (Q2..Q10). */
int vbxo_pitch(const double *x, size_t n, double sample_rate, double threshold,
               double fmin, double fmax, vbxo_pitch_t *out, size_t cap, size_t *count) {
    double *window_lag = (double *)malloc(n * sizeof(double));
    double *self_lag = (double *)calloc(2 * n, sizeof(double));   /* :411 resize(2N, 0) */
    vbxo_window_hanning_lag(window_lag, n);                                        /* :400 */
    vbxo_autocorrelate(x, n, self_lag, n);                                         /* :403 */
    vbxo_normalize(self_lag, n);                                                   /* :404 */
    for (size_t i = 0; i < n; i++) self_lag[i] = self_lag[i] / window_lag[i];      /* :406-408 */
    int status = pitch_from_lag(self_lag, n, sample_rate, threshold, fmin, fmax, out, cap, count, 0);
    free(window_lag); free(self_lag);
    return status;
}

/* The S = T = f32 instantiation (parity unpinned): the lag curve in f32 arithmetic (vbx_oracle_f32.c), the lag window
 * as the f64 table rounded to f32, then the rest on the widened curve with T = f32 roundings. */
void vbxo_autocorrelate_f32(const float *x, size_t n, float *coeffs, size_t n_lags);
void vbxo_normalize_f32(float *x, size_t n);
int vbxo_pitch_f32(const float *x, size_t n, float sample_rate, float threshold, float fmin, float fmax,
                   vbxo_pitch_t *out, size_t cap, size_t *count) {
    double *window_lag = (double *)malloc(n * sizeof(double));
    float *lag32 = (float *)malloc(n * sizeof(float));
    double *self_lag = (double *)calloc(2 * n, sizeof(double));
    vbxo_window_hanning_lag(window_lag, n);
    vbxo_autocorrelate_f32(x, n, lag32, n);
    vbxo_normalize_f32(lag32, n);
    for (size_t i = 0; i < n; i++) self_lag[i] = (double)(lag32[i] / (float)window_lag[i]);
    int status = pitch_from_lag(self_lag, n, (double)sample_rate, (double)threshold, (double)fmin, (double)fmax, out, cap, count, 1);
    free(window_lag); free(lag32); free(self_lag);
    return status;
}

/* ------------------------------------------------------------------------ */
/* spectrum.rs: LPC                                                          */
/* ------------------------------------------------------------------------ */

/* spectrum.rs:63-84 (Levinson-Durbin) through lpc() :86-92 (zeroed work vectors). */
void vbxo_lpc(const double *r, size_t n_coeffs, double *ac) {
    double *kc = (double *)calloc(n_coeffs ? n_coeffs : 1, sizeof(double));
    double *tmp = (double *)calloc(n_coeffs ? n_coeffs : 1, sizeof(double));
    for (size_t i = 0; i < n_coeffs + 1; i++) ac[i] = 0.0;
    double err = r[0];
    ac[0] = 1.0;
    for (size_t i = 1; i < n_coeffs + 1; i++) {
        double acc = r[i];
        for (size_t j = 1; j < i; j++) acc = acc + (ac[j] * r[i - j]);
        kc[i - 1] = -acc / err;
        ac[i] = kc[i - 1];
        for (size_t j = 0; j < n_coeffs; j++) tmp[j] = ac[j];
        for (size_t j = 1; j < i; j++) ac[j] = ac[j] + (kc[i - 1] * tmp[i - j]);
        err = err * (1.0 - (kc[i - 1] * kc[i - 1]));
    }
    free(kc); free(tmp);
}

/* spectrum.rs:101-146 (Burg, Q12).  coeffs has n_coeffs entries, no leading 1. */
int vbxo_lpc_burg(const double *x, size_t n, size_t n_coeffs, double *coeffs) {
    if (n < 2) return VBXO_ERR_PANIC;  /* self.len() - 2 underflows / OOB in the reference */
    double *b1 = (double *)calloc(n, sizeof(double));
    double *b2 = (double *)calloc(n, sizeof(double));
    double *aa = (double *)calloc(n_coeffs ? n_coeffs : 1, sizeof(double));
    int status = VBXO_OK;

    b1[0] = x[0];
    b2[n - 2] = x[n - 1];
    for (size_t j = 2; j < n; j++) {
        b1[j - 1] = x[j - 1];
        b2[j - 2] = x[j - 1];
    }

    for (size_t i = 1; i < n_coeffs + 1; i++) {
        double num = 0.0, denum = 0.0;
        for (size_t j = 1; j + i < n + 1; j++) {       /* j in 1 .. len - i + 1 */
            num = num + b1[j - 1] * b2[j - 1];
            denum = denum + b1[j - 1] * b1[j - 1] + b2[j - 1] * b2[j - 1];
        }
        if (denum <= 0.0) { status = VBXO_ERR_LPC; break; }   /* NaN: comparison false, continues */
        coeffs[i - 1] = 2.0 * num / denum;
        for (size_t j = 1; j < i; j++) coeffs[j - 1] = aa[j - 1] - coeffs[i - 1] * aa[i - j - 1];
        if (i < n_coeffs) {
            for (size_t j = 1; j < i + 1; j++) aa[j - 1] = coeffs[j - 1];
            for (size_t j = 1; j + i < n; j++) {       /* j in 1 .. len - i */
                b1[j - 1] = b1[j - 1] - aa[i - 1] * b2[j - 1];
                b2[j - 1] = b2[j] - aa[i - 1] * b1[j];
            }
        }
    }
    if (status == VBXO_OK)
        for (size_t c = 0; c < n_coeffs; c++) coeffs[c] = coeffs[c] * -1.0;
    free(b1); free(b2); free(aa);
    return status;
}

/* ------------------------------------------------------------------------ */
/* num-complex 0.2 arithmetic (not in /root/reference)                       */
/* ------------------------------------------------------------------------ */

static vbxo_c64 c_new(double re, double im) { vbxo_c64 z = { re, im }; return z; }
static vbxo_c64 c_add(vbxo_c64 a, vbxo_c64 b) { return c_new(a.re + b.re, a.im + b.im); }
static vbxo_c64 c_sub(vbxo_c64 a, vbxo_c64 b) { return c_new(a.re - b.re, a.im - b.im); }
static vbxo_c64 c_neg(vbxo_c64 a) { return c_new(-a.re, -a.im); }
static vbxo_c64 c_mul(vbxo_c64 a, vbxo_c64 b) {
    return c_new(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
static vbxo_c64 c_div(vbxo_c64 a, vbxo_c64 b) {
    double norm_sqr = b.re * b.re + b.im * b.im;
    double re = a.re * b.re + a.im * b.im;
    double im = a.im * b.re - a.re * b.im;
    return c_new(re / norm_sqr, im / norm_sqr);
}
static double c_norm(vbxo_c64 a) { return hypot(a.re, a.im); }
static int c_is_zero(vbxo_c64 a) { return a.re == 0.0 && a.im == 0.0; }
/* Complex::sqrt (num-complex 0.2.4): purely real / purely imaginary inputs are
 * special-cased, everything else goes through polar form. */
static vbxo_c64 c_sqrt(vbxo_c64 z) {
    if (z.im == 0.0) {
        if (!signbit(z.re)) return c_new(sqrt(z.re), z.im);
        double re = 0.0, im = sqrt(-z.re);
        return signbit(z.im) ? c_new(re, -im) : c_new(re, im);
    } else if (z.re == 0.0) {
        double x = sqrt(fabs(z.im) / 2.0);
        return signbit(z.im) ? c_new(x, -x) : c_new(x, x);
    }
    double r = hypot(z.re, z.im), theta = atan2(z.im, z.re);
    double sr = sqrt(r), th = theta / 2.0;
    return c_new(sr * cos(th), sr * sin(th));
}

/* ------------------------------------------------------------------------ */
/* polynomial.rs                                                             */
/* ------------------------------------------------------------------------ */

/* polynomial.rs:26-28 */
size_t vbxo_degree(const vbxo_c64 *p, size_t len) {
    for (size_t i = len; i-- > 0;) if (!c_is_zero(p[i])) return i;
    return 0;
}
/* polynomial.rs:30-32 */
size_t vbxo_off_low(const vbxo_c64 *p, size_t len) {
    for (size_t i = 0; i < len; i++) if (!c_is_zero(p[i])) return i;
    return 0;
}

/* polynomial.rs:34-72 (Q11: n = len - 1 fixed; radicand (n-1)*n*H - G^2). */
vbxo_c64 vbxo_laguerre(const vbxo_c64 *p, size_t len, vbxo_c64 start) {
    size_t n = len - 1;
    vbxo_c64 z = start;
    for (int it = 0; it < 20; it++) {
        vbxo_c64 abg0 = p[n], abg1 = c_new(0, 0), abg2 = c_new(0, 0);
        for (size_t j = n; j-- > 0;) {
            abg2 = c_add(c_mul(abg2, z), abg1);
            abg1 = c_add(c_mul(abg1, z), abg0);
            abg0 = c_add(c_mul(abg0, z), p[j]);
        }
        if (c_norm(abg0) <= 1.0e-16) return z;
        vbxo_c64 ca = c_div(c_neg(abg1), abg0);
        vbxo_c64 ca2 = c_mul(ca, ca);
        vbxo_c64 cb = c_sub(ca2, c_div(c_mul(c_new(2.0, 0.0), abg2), abg0));
        vbxo_c64 c1 = c_sqrt(c_sub(c_mul(c_mul(c_new((double)(n - 1), 0.0), c_new((double)n, 0.0)), cb), ca2));
        vbxo_c64 cc1 = c_add(ca, c1);
        vbxo_c64 cc2 = c_sub(ca, c1);
        vbxo_c64 cc = (c_norm(cc1) > c_norm(cc2)) ? c_div(c_new((double)n, 0.0), cc1)
                                                  : c_div(c_new((double)n, 0.0), cc2);
        z = c_add(z, cc);
    }
    return z;
}

/* polynomial.rs:155-195, `other` != 0 branch (ds = 1). */
static int div_polynomial_mut(vbxo_c64 *self, size_t len, vbxo_c64 other, vbxo_c64 *rem) {
    for (size_t i = 0; i < len; i++) rem[i] = self[i];
    if (c_is_zero(other)) return VBXO_ERR_POLYNOMIAL;                               /* :192 */
    size_t ns = vbxo_degree(self, len);
    const size_t ds = 1;
    for (size_t i = ns - ds + 1; i-- > 0;) {
        self[i] = rem[ds + i];
        rem[i] = c_sub(rem[i], c_mul(self[i], other));      /* j == i only (ds == 1) */
    }
    for (size_t k = ds; k < ns + 1; k++) rem[vbxo_degree(rem, len)] = c_new(0, 0);  /* :174-176 */
    size_t l = vbxo_degree(self, len);
    size_t cnt = (l + 1) - ns - ds + 1;                                              /* :179 */
    for (size_t k = 0; k < cnt; k++) self[vbxo_degree(self, len)] = c_new(0, 0);
    return VBXO_OK;
}
/* the public trait method (polynomial.rs:155): quotient of self / (x + other) left in self, remainder in rem */
int vbxo_div_polynomial_mut(vbxo_c64 *self, size_t len, vbxo_c64 other, vbxo_c64 *rem) {
    return div_polynomial_mut(self, len, other, rem);
}

/* polynomial.rs:92-152.  The caller's work slice is restated as zeroed scratch
 * (as in find_roots :80 and tests/lib.rs:34,67). */
int vbxo_find_roots_mut(vbxo_c64 *self, size_t len) {
    size_t coeff_high = vbxo_degree(self, len);
    if (coeff_high < 1) return VBXO_ERR_POLYNOMIAL;                                  /* :95 */
    size_t coeff_low = vbxo_off_low(self, len);
    size_t m = coeff_high - coeff_low;
    size_t clen = coeff_high - coeff_low + 1;
    if (coeff_high >= clen) return VBXO_ERR_PANIC;        /* :110-112 OOB when off_low > 0 */

    vbxo_c64 *z_roots = (vbxo_c64 *)calloc(2 * len, sizeof(vbxo_c64));
    vbxo_c64 *rem = (vbxo_c64 *)calloc(clen, sizeof(vbxo_c64));
    vbxo_c64 *coeffs = (vbxo_c64 *)calloc(clen, sizeof(vbxo_c64));
    size_t z_root_index = 0;
    int status = VBXO_OK;
    for (size_t i = 0; i < coeff_low; i++) { z_roots[i] = c_new(0, 0); z_root_index++; }
    for (size_t co = coeff_low; co < coeff_high + 1; co++) coeffs[co] = self[co];

    size_t m0 = m;
    for (size_t k = m0 + 1; k-- > 3;) {                   /* (3..m+1).rev(): m-2 passes */
        vbxo_c64 z = vbxo_laguerre(coeffs, clen, c_new(-2.0, -2.0));                 /* :117-118 */
        z_roots[z_root_index++] = z;
        if (div_polynomial_mut(coeffs, clen, c_neg(z), rem) != VBXO_OK) { status = VBXO_ERR_POLYNOMIAL; break; }
        m = m - 1;
    }
    if (status == VBXO_OK) {
        if (m == 2) {                                                                /* :131-139 */
            vbxo_c64 a2 = c_add(coeffs[2], coeffs[2]);
            vbxo_c64 d = c_sqrt(c_sub(c_mul(coeffs[1], coeffs[1]),
                                      c_mul(c_mul(c_new(4.0, 0.0), coeffs[2]), coeffs[0])));
            vbxo_c64 x = c_neg(coeffs[1]);
            z_roots[z_root_index] = c_div(c_add(x, d), a2);
            z_roots[z_root_index + 1] = c_div(c_sub(x, d), a2);
            z_root_index += 2;
        }
        if (m == 1) {                                                                /* :141-144 */
            z_roots[z_root_index] = c_div(c_neg(coeffs[0]), coeffs[1]);
            z_root_index += 1;
        }
        if (z_root_index + 1 > len) status = VBXO_ERR_PANIC;
        else {
            for (size_t i = 0; i < z_root_index + 1; i++) self[i] = z_roots[i];      /* :145-147 */
            for (size_t i = z_root_index + 1; i < len; i++) self[i] = c_new(0, 0);   /* :148-150 */
        }
    }
    free(z_roots); free(rem); free(coeffs);
    return status;
}

/* polynomial.rs:79-89 */
int vbxo_find_roots(const vbxo_c64 *p, size_t len, vbxo_c64 *roots, size_t *n_roots) {
    vbxo_c64 *other = (vbxo_c64 *)malloc(len * sizeof(vbxo_c64));
    memcpy(other, p, len * sizeof(vbxo_c64));
    int st = vbxo_find_roots_mut(other, len);
    size_t l = len;
    if (st == VBXO_OK) {
        while (l > 0 && c_is_zero(other[l - 1])) l--;    /* reference panics at l == 0 */
        memcpy(roots, other, l * sizeof(vbxo_c64));
        *n_roots = l;
    } else *n_roots = 0;
    free(other);
    return st;
}

/* ------------------------------------------------------------------------ */
/* spectrum.rs: resonances and the formant tracker                           */
/* ------------------------------------------------------------------------ */

/* spectrum.rs:166-192 */
int vbxo_resonance_from_root(vbxo_c64 root, double sample_rate, vbxo_resonance_t *out) {
    double freq_mul = sample_rate / (M_PI * 2.0);
    if (root.im >= 0.0) {
        double r = hypot(root.re, root.im), theta = atan2(root.im, root.re);
        if (r > 1.0) {
            /* root.conj().inv(): conj = (re, -im); inv = (re/ns, -im/ns) */
            double cre = root.re, cim = -root.im;
            double ns = cre * cre + cim * cim;
            double ire = cre / ns, iim = -cim / ns;
            r = hypot(ire, iim); theta = atan2(iim, ire);
        }
        double frequency = freq_mul * theta;
        double bandwidth = -2.0 * freq_mul * log(r);
        double safety = 50.0, nyquist = sample_rate * 0.5;
        if (frequency > safety && frequency < nyquist - safety) {
            out->frequency = frequency; out->bandwidth = bandwidth;
            return 1;
        }
    }
    return 0;
}

/* spectrum.rs:204-209 (stable sort by frequency) */
size_t vbxo_to_resonance(const vbxo_c64 *roots, size_t n, double sample_rate, vbxo_resonance_t *out) {
    size_t cnt = 0;
    for (size_t i = 0; i < n; i++) if (vbxo_resonance_from_root(roots[i], sample_rate, &out[cnt])) cnt++;
    for (size_t i = 1; i < cnt; i++) {
        vbxo_resonance_t key = out[i]; size_t j = i;
        while (j > 0 && out[j - 1].frequency > key.frequency) { out[j] = out[j - 1]; j--; }
        out[j] = key;
    }
    return cnt;
}

typedef struct { int some; vbxo_resonance_t v; } opt_res;
static int res_eq(vbxo_resonance_t a, vbxo_resonance_t b) {   /* derive(PartialEq), :149 */
    return a.frequency == b.frequency && a.bandwidth == b.bandwidth;
}
static int opt_eq(opt_res a, opt_res b) {
    if (a.some != b.some) return 0;
    return a.some ? res_eq(a.v, b.v) : 1;
}
static int slots_contains(const opt_res *slots, opt_res peak) {
    for (int i = 0; i < VBXO_FORMANT_SLOTS; i++) if (opt_eq(slots[i], peak)) return 1;
    return 0;
}
/* comparator of :312-324: returns <0, 0, >0 */
static int slot_cmp(opt_res a, opt_res b) {
    if (a.some) {
        if (b.some) {
            if (a.v.frequency < b.v.frequency) return -1;
            if (a.v.frequency > b.v.frequency) return 1;
            return 0;                                   /* Equal, also for NaN (unwrap_or) */
        }
        return 1;                                       /* Greater */
    }
    return -1;                                          /* Less */
}

/* spectrum.rs:232-333 (Q13). */
void vbxo_estimate_formants(vbxo_resonance_t *self, size_t n_est,
                            const vbxo_resonance_t *resonances, size_t n_res) {
    opt_res slots[VBXO_FORMANT_SLOTS];
    memset(slots, 0, sizeof slots);
    size_t n_zip = n_est < VBXO_FORMANT_SLOTS ? n_est : VBXO_FORMANT_SLOTS;

    /* Step 2 (:235-245): nearest resonance (strict <, first wins ties) per estimate */
    for (size_t e = 0; e < n_zip; e++) {
        vbxo_resonance_t best = resonances[0];
        double bestd = fabs(resonances[0].frequency - self[e].frequency);
        for (size_t i = 1; i < n_res; i++) {
            double d = fabs(resonances[i].frequency - self[e].frequency);
            if (d < bestd) { best = resonances[i]; bestd = d; }
        }
        slots[e].some = 1; slots[e].v = best;
    }

    /* Step 3 (:250-272) */
    size_t w = 0;
    int has_unassigned = 0;
    for (size_t r = 1; r < VBXO_FORMANT_SLOTS; r++) {
        if (!slots[r].some) continue;
        vbxo_resonance_t v = slots[r].v;
        if (res_eq(v, slots[w].v)) {
            if (fabs(v.frequency - self[r].frequency) < fabs(v.frequency - self[w].frequency)) {
                slots[w].some = 0; has_unassigned = 1; w = r;
            } else {
                slots[r].some = 0; has_unassigned = 1;
            }
        } else {
            w = r;
        }
    }

    /* Step 4 (:274-310) */
    if (has_unassigned) {
        for (size_t j = 0; j < n_res; j++) {
            opt_res peak; peak.some = 1; peak.v = resonances[j];
            if (slots_contains(slots, peak)) continue;
            if (j < VBXO_FORMANT_SLOTS && !slots[j].some) { slots[j] = peak; continue; }
            if (j > 0 && j < VBXO_FORMANT_SLOTS) {
                if (!slots[j - 1].some) {
                    opt_res t = slots[j]; slots[j] = slots[j - 1]; slots[j - 1] = t;
                    slots[j] = peak; continue;
                }
            }
            if (j + 1 < VBXO_FORMANT_SLOTS && !slots[j + 1].some) {
                opt_res t = slots[j]; slots[j] = slots[j + 1]; slots[j + 1] = t;
                slots[j] = peak; continue;
            }
        }
    }

    /* :312-324 stable sort (None first, then ascending frequency) */
    for (int i = 1; i < VBXO_FORMANT_SLOTS; i++) {
        opt_res key = slots[i]; int j = i;
        while (j > 0 && slot_cmp(slots[j - 1], key) > 0) { slots[j] = slots[j - 1]; j--; }
        slots[j] = key;
    }

    /* :327-332 winners with frequency > 0 overwrite the leading estimates */
    size_t e = 0;
    for (int i = 0; i < VBXO_FORMANT_SLOTS && e < n_est; i++) {
        if (slots[i].some && slots[i].v.frequency > 0.0) self[e++] = slots[i].v;
    }
}

/* ------------------------------------------------------------------------ */
/* lib.rs: find_formants (resample_ratio == 1.0)                             */
/* ------------------------------------------------------------------------ */

int vbxo_find_formants(const double *x, size_t n, double sample_rate, size_t n_coeffs,
                       vbxo_resonance_t *formants, size_t n_formants,
                       vbxo_resonance_t *res_out, double *coeffs_out) {
    vbxo_resonance_t resonances[VBXO_MAX_RESONANCES];
    memset(resonances, 0, sizeof resonances);                                        /* :55 */
    double *buf = (double *)malloc(n * sizeof(double));
    double *lpc = (double *)calloc(n_coeffs ? n_coeffs : 1, sizeof(double));
    vbxo_c64 *cl = (vbxo_c64 *)calloc(n_coeffs + 1, sizeof(vbxo_c64));
    int status;

    /* :63 copy, :65-70 periodic Hanning */
    double len_inv = 1.0 / (double)n;
    for (size_t idx = 0; idx < n; idx++) {
        double window = 0.5 * (1.0 - cos(((double)idx * len_inv) * (M_PI * 2.0)));
        buf[idx] = x[idx] * window;
    }
    status = vbxo_lpc_burg(buf, n, n_coeffs, lpc);                                   /* :75 */
    if (status == VBXO_OK) {
        if (coeffs_out) memcpy(coeffs_out, lpc, n_coeffs * sizeof(double));
        /* :80-91 complex_lpc = rev([1, a1..ap]) */
        for (size_t i = 0; i < n_coeffs + 1; i++) {
            double r = (i < n_coeffs) ? lpc[n_coeffs - 1 - i] : 1.0;
            cl[i].re = r; cl[i].im = 0.0;
        }
        status = vbxo_find_roots_mut(cl, n_coeffs + 1);                              /* :93 */
    }
    if (status == VBXO_OK) {
        size_t count = 0;
        for (size_t i = 0; i < n_coeffs + 1; i++) {                                  /* :94-104 */
            if (cl[i].im > 0.0) {
                vbxo_resonance_t r;
                if (vbxo_resonance_from_root(cl[i], sample_rate, &r)) {
                    if (count >= VBXO_MAX_RESONANCES) { status = VBXO_ERR_PANIC; break; }
                    resonances[count++] = r;
                }
            }
        }
    }
    if (status == VBXO_OK) {
        size_t rpos = 0;                                                             /* :105-107 */
        for (size_t i = VBXO_MAX_RESONANCES; i-- > 0;) if (resonances[i].frequency != 0.0) { rpos = i; break; }
        for (size_t i = 1; i < rpos + 1; i++) {                                      /* :108-110 stable */
            vbxo_resonance_t key = resonances[i]; size_t j = i;
            while (j > 0 && resonances[j - 1].frequency > key.frequency) { resonances[j] = resonances[j - 1]; j--; }
            resonances[j] = key;
        }
        if (res_out) memcpy(res_out, resonances, sizeof resonances);
        vbxo_estimate_formants(formants, n_formants, resonances, VBXO_MAX_RESONANCES); /* :114 */
    }
    free(buf); free(lpc); free(cl);
    return status;
}

/* lib.rs:42,57-61 (resample_ratio != 1.0); sample 0.10 Converter + Linear restated -- parity unpinned. */
size_t vbxo_resampled_len(size_t n, double ratio) { return (size_t)ceil(ratio * (double)n); }

void vbxo_resample_linear(const double *x, size_t n, double ratio, double *out) {
    size_t m = vbxo_resampled_len(n, ratio);
    size_t next = 0;                                   /* signal::from_iter: equilibrium after the end */
    double left = (next < n) ? x[next] : 0.0; next++;  /* Linear::new(buf_iter.next(), buf_iter.next()) */
    double right = (next < n) ? x[next] : 0.0; next++;
    double interpolation_value = 0.0;
    double source_to_target_ratio = 1.0 / ratio;       /* scale_sample_hz(.., scale) = scale_playback_hz(.., 1/scale) */
    for (size_t k = 0; k < m; k++) {
        while (interpolation_value >= 1.0) {
            left = right;
            right = (next < n) ? x[next] : 0.0; next++;
            interpolation_value -= 1.0;
        }
        double diff = right - left;
        out[k] = (diff * interpolation_value) + left;
        interpolation_value += source_to_target_ratio;
    }
}

int vbxo_find_formants_ratio(const double *x, size_t n, double sample_rate, double ratio, size_t n_coeffs,
                             vbxo_resonance_t *formants, size_t n_formants) {
    if (ratio == 1.0) return vbxo_find_formants(x, n, sample_rate, n_coeffs, formants, n_formants, NULL, NULL);
    size_t m = vbxo_resampled_len(n, ratio);
    double *buf = (double *)malloc((m ? m : 1) * sizeof(double));
    vbxo_resample_linear(x, n, ratio, buf);
    int st = vbxo_find_formants(buf, m, sample_rate, n_coeffs, formants, n_formants, NULL, NULL);
    free(buf);
    return st;
}

/* ------------------------------------------------------------------------ */
/* spectrum.rs: MFCC                                                         */
/* ------------------------------------------------------------------------ */

double vbxo_hz_to_mel(double hz) { return 1125. * log1p(hz / 700.); }               /* :375-377 */
double vbxo_mel_to_hz(double mel) { return 700. * (exp(mel / 1125.) - 1.); }         /* :379-381 */

/* spectrum.rs:391-398 */
void vbxo_dct(const double *signal, size_t n, double *coeffs) {
    for (size_t k = 0; k < n; k++) {
        double acc = 0.;
        for (size_t i = 0; i < n; i++)
            acc = acc + signal[i] * cos(M_PI * (double)k * (2. * (double)i + 1.) / (2. * (double)n));
        coeffs[k] = 2. * acc;
    }
}

/* spectrum.rs:411-414 (Q14: two points beyond hi; bins scale with N+1) */
void vbxo_mfcc_bins(size_t n, size_t num_coeffs, double lo, double hi, double sr, size_t *bins) {
    double mel_range = vbxo_hz_to_mel(hi) - vbxo_hz_to_mel(lo);
    for (size_t i = 0; i < num_coeffs + 2; i++) {
        double point = ((double)i / (double)num_coeffs) * mel_range + vbxo_hz_to_mel(lo);
        double b = floor((double)(n + 1) * vbxo_mel_to_hz(point) / sr);
        bins[i] = (b > 0.0) ? (size_t)b : 0;
    }
}

/* Mixed-radix decimation-in-time FFT (any n; generic O(r^2) butterflies for
 * prime radices).  Stands in for rustfft 1.0's FFT::new(n, false).process. */
static void fft_rec(const vbxo_c64 *in, size_t stride, vbxo_c64 *out, size_t n,
                    const vbxo_c64 *tw, size_t tw_stride, vbxo_c64 *scratch) {
    if (n == 1) { out[0] = in[0]; return; }
    size_t r = n;
    for (size_t f = 2; f * f <= n; f++) if (n % f == 0) { r = f; break; }
    size_t m = n / r;
    for (size_t q = 0; q < r; q++)
        fft_rec(in + q * stride, stride * r, out + q * m, m, tw, tw_stride * r, scratch);
    /* combine: X[k + m*s] = sum_q W_n^{q(k+m*s)} Y_q[k] */
    size_t big = n * tw_stride;   /* = N of the root transform */
    for (size_t k = 0; k < m; k++) {
        for (size_t q = 0; q < r; q++) scratch[q] = out[q * m + k];
        for (size_t s = 0; s < r; s++) {
            size_t kk = k + m * s;
            vbxo_c64 acc = scratch[0];
            for (size_t q = 1; q < r; q++) {
                size_t ti = ((q * kk) % n) * tw_stride;
                acc = c_add(acc, c_mul(scratch[q], tw[ti % big]));
            }
            out[kk] = acc;   /* safe: the r outputs for this k only overwrite slots q*m+k */
        }
    }
}

void vbxo_fft(const vbxo_c64 *in, vbxo_c64 *out, size_t n) {
    vbxo_c64 *tw = (vbxo_c64 *)malloc(n * sizeof(vbxo_c64));
    vbxo_c64 scratch[64];
    vbxo_c64 *heap_scratch = NULL, *sc = scratch;
    for (size_t i = 0; i < n; i++) {
        double ang = -2.0 * M_PI * (double)i / (double)n;
        tw[i] = c_new(cos(ang), sin(ang));
    }
    if (n > 64) { heap_scratch = (vbxo_c64 *)malloc(n * sizeof(vbxo_c64)); sc = heap_scratch; }
    fft_rec(in, 1, out, n, tw, 1, sc);
    free(tw); free(heap_scratch);
}

/* spectrum.rs:410-440 (Q14). */
int vbxo_mfcc(const double *x, size_t n, size_t num_coeffs, double lo, double hi,
              double sample_rate, double *out, int use_fft) {
    size_t *bins = (size_t *)malloc((num_coeffs + 2) * sizeof(size_t));
    vbxo_mfcc_bins(n, num_coeffs, lo, hi, sample_rate, bins);
    size_t top = bins[num_coeffs + 1];
    if (top > n) { free(bins); return VBXO_ERR_PANIC; }   /* spectrum[bin] OOB */
    vbxo_c64 *spectrum = (vbxo_c64 *)calloc(n, sizeof(vbxo_c64));
    if (use_fft) {
        vbxo_c64 *sig = (vbxo_c64 *)malloc(n * sizeof(vbxo_c64));
        for (size_t i = 0; i < n; i++) sig[i] = c_new(x[i], 0.0);
        vbxo_fft(sig, spectrum, n);
        free(sig);
    } else {
        double *ct = (double *)malloc(n * sizeof(double)), *st = (double *)malloc(n * sizeof(double));
        for (size_t i = 0; i < n; i++) {
            double ang = 2.0 * M_PI * (double)i / (double)n;
            ct[i] = cos(ang); st[i] = sin(ang);
        }
        for (size_t k = bins[0]; k < top; k++) {
            double re = 0., im = 0.;
            size_t idx = 0;
            for (size_t i = 0; i < n; i++) {
                re += x[i] * ct[idx]; im -= x[i] * st[idx];
                idx += k; if (idx >= n) idx -= n;
            }
            spectrum[k] = c_new(re, im);
        }
        free(ct); free(st);
    }
    double *energies = (double *)malloc(num_coeffs * sizeof(double));
    for (size_t wdx = 0; wdx < num_coeffs; wdx++) {                                  /* :421-437 */
        size_t w0 = bins[wdx], w1 = bins[wdx + 1], w2 = bins[wdx + 2];
        size_t up = w1 - w0;
        double up_sum = 0.;
        for (size_t i = 0, bin = w0; bin < w1; i++, bin++) {
            double multiplier = (double)i / (double)up;
            double ns = spectrum[bin].re * spectrum[bin].re + spectrum[bin].im * spectrum[bin].im;
            up_sum = up_sum + fabs(ns) * multiplier;
        }
        size_t down = w2 - w1;
        double down_sum = 0.;
        for (size_t i = 0, bin = w1; bin < w2; i++, bin++) {
            double multiplier = (double)i / (double)down;
            down_sum = down_sum + fabs(hypot(spectrum[bin].re, spectrum[bin].im)) * multiplier;
        }
        double lg = log10(up_sum + down_sum);
        /* f64::max: NaN operand yields the other one */
        energies[wdx] = (isnan(lg) || lg < 1.0e-10) ? 1.0e-10 : lg;
    }
    vbxo_dct(energies, num_coeffs, out);                                             /* :439 */
    free(bins); free(spectrum); free(energies);
    return VBXO_OK;
}
