/*
 * vbx_oracle_f32.c -- the Complex<f32> instantiation of the reference's Polynomial trait
 * (src/polynomial.rs:10-205), restated in plain C single precision.
 *
 * TEST INFRASTRUCTURE ONLY (see vbx_oracle.h).  Pinned by the reference's own f32 tests,
 * src/polynomial.rs:336-386 (test_2d_complex_roots_f32 to 1e-12, test_hi_d_roots_f32 to 1e-6,
 * test_f32_roots finiteness) -- tests/test_oracle_kat.py.
 * num-complex 0.2 arithmetic (mul/div/norm/sqrt) restated as in vbx_oracle.c, with the f32 libm
 * (hypotf, atan2f, sqrtf, cosf, sinf) where Rust's f32 methods call theirs.
 */
#include "vbx_oracle.h"

#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdlib.h>
#include <string.h>

static vbxo_c32 c_new(float re, float im) { vbxo_c32 z = { re, im }; return z; }
static vbxo_c32 c_add(vbxo_c32 a, vbxo_c32 b) { return c_new(a.re + b.re, a.im + b.im); }
static vbxo_c32 c_sub(vbxo_c32 a, vbxo_c32 b) { return c_new(a.re - b.re, a.im - b.im); }
static vbxo_c32 c_neg(vbxo_c32 a) { return c_new(-a.re, -a.im); }
static vbxo_c32 c_mul(vbxo_c32 a, vbxo_c32 b) {
    return c_new(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
static vbxo_c32 c_div(vbxo_c32 a, vbxo_c32 b) {
    float norm_sqr = b.re * b.re + b.im * b.im;
    float re = a.re * b.re + a.im * b.im;
    float im = a.im * b.re - a.re * b.im;
    return c_new(re / norm_sqr, im / norm_sqr);
}
static float c_norm(vbxo_c32 a) { return hypotf(a.re, a.im); }
static int c_is_zero(vbxo_c32 a) { return a.re == 0.0f && a.im == 0.0f; }
/* Complex::sqrt (num-complex 0.2.4) */
static vbxo_c32 c_sqrt(vbxo_c32 z) {
    if (z.im == 0.0f) {
        if (!signbit(z.re)) return c_new(sqrtf(z.re), z.im);
        float re = 0.0f, im = sqrtf(-z.re);
        return signbit(z.im) ? c_new(re, -im) : c_new(re, im);
    } else if (z.re == 0.0f) {
        float x = sqrtf(fabsf(z.im) / 2.0f);
        return signbit(z.im) ? c_new(x, -x) : c_new(x, x);
    }
    float r = hypotf(z.re, z.im), theta = atan2f(z.im, z.re);
    float sr = sqrtf(r), th = theta / 2.0f;
    return c_new(sr * cosf(th), sr * sinf(th));
}

static size_t degree32(const vbxo_c32 *p, size_t len) {                     /* polynomial.rs:26-28 */
    for (size_t i = len; i-- > 0;) if (!c_is_zero(p[i])) return i;
    return 0;
}
static size_t off_low32(const vbxo_c32 *p, size_t len) {                    /* polynomial.rs:30-32 */
    for (size_t i = 0; i < len; i++) if (!c_is_zero(p[i])) return i;
    return 0;
}

/* polynomial.rs:34-72 */
vbxo_c32 vbxo_laguerre_f32(const vbxo_c32 *p, size_t len, vbxo_c32 start) {
    size_t n = len - 1;
    vbxo_c32 z = start;
    for (int it = 0; it < 20; it++) {
        vbxo_c32 abg0 = p[n], abg1 = c_new(0, 0), abg2 = c_new(0, 0);
        for (size_t j = n; j-- > 0;) {
            abg2 = c_add(c_mul(abg2, z), abg1);
            abg1 = c_add(c_mul(abg1, z), abg0);
            abg0 = c_add(c_mul(abg0, z), p[j]);
        }
        if (c_norm(abg0) <= 1.0e-16f) return z;
        vbxo_c32 ca = c_div(c_neg(abg1), abg0);
        vbxo_c32 ca2 = c_mul(ca, ca);
        vbxo_c32 cb = c_sub(ca2, c_div(c_mul(c_new(2.0f, 0.0f), abg2), abg0));
        vbxo_c32 c1 = c_sqrt(c_sub(c_mul(c_mul(c_new((float)(n - 1), 0.0f), c_new((float)n, 0.0f)), cb), ca2));
        vbxo_c32 cc1 = c_add(ca, c1);
        vbxo_c32 cc2 = c_sub(ca, c1);
        vbxo_c32 cc = (c_norm(cc1) > c_norm(cc2)) ? c_div(c_new((float)n, 0.0f), cc1)
                                                  : c_div(c_new((float)n, 0.0f), cc2);
        z = c_add(z, cc);
    }
    return z;
}

/* polynomial.rs:155-195, `other` != 0 branch (ds = 1) */
static int div_polynomial_mut32(vbxo_c32 *self, size_t len, vbxo_c32 other, vbxo_c32 *rem) {
    for (size_t i = 0; i < len; i++) rem[i] = self[i];
    if (c_is_zero(other)) return VBXO_ERR_POLYNOMIAL;
    size_t ns = degree32(self, len);
    const size_t ds = 1;
    for (size_t i = ns - ds + 1; i-- > 0;) {
        self[i] = rem[ds + i];
        rem[i] = c_sub(rem[i], c_mul(self[i], other));
    }
    for (size_t k = ds; k < ns + 1; k++) rem[degree32(rem, len)] = c_new(0, 0);
    size_t l = degree32(self, len);
    size_t cnt = (l + 1) - ns - ds + 1;
    for (size_t k = 0; k < cnt; k++) self[degree32(self, len)] = c_new(0, 0);
    return VBXO_OK;
}

/* polynomial.rs:92-152 */
int vbxo_find_roots_mut_f32(vbxo_c32 *self, size_t len) {
    size_t coeff_high = degree32(self, len);
    if (coeff_high < 1) return VBXO_ERR_POLYNOMIAL;
    size_t coeff_low = off_low32(self, len);
    size_t m = coeff_high - coeff_low;
    size_t clen = coeff_high - coeff_low + 1;
    if (coeff_high >= clen) return VBXO_ERR_PANIC;

    vbxo_c32 *z_roots = (vbxo_c32 *)calloc(2 * len, sizeof(vbxo_c32));
    vbxo_c32 *rem = (vbxo_c32 *)calloc(clen, sizeof(vbxo_c32));
    vbxo_c32 *coeffs = (vbxo_c32 *)calloc(clen, sizeof(vbxo_c32));
    size_t z_root_index = 0;
    int status = VBXO_OK;
    for (size_t i = 0; i < coeff_low; i++) { z_roots[i] = c_new(0, 0); z_root_index++; }
    for (size_t co = coeff_low; co < coeff_high + 1; co++) coeffs[co] = self[co];

    size_t m0 = m;
    for (size_t k = m0 + 1; k-- > 3;) {
        vbxo_c32 z = vbxo_laguerre_f32(coeffs, clen, c_new(-2.0f, -2.0f));
        z_roots[z_root_index++] = z;
        if (div_polynomial_mut32(coeffs, clen, c_neg(z), rem) != VBXO_OK) { status = VBXO_ERR_POLYNOMIAL; break; }
        m = m - 1;
    }
    if (status == VBXO_OK) {
        if (m == 2) {
            vbxo_c32 a2 = c_add(coeffs[2], coeffs[2]);
            vbxo_c32 d = c_sqrt(c_sub(c_mul(coeffs[1], coeffs[1]),
                                      c_mul(c_mul(c_new(4.0f, 0.0f), coeffs[2]), coeffs[0])));
            vbxo_c32 x = c_neg(coeffs[1]);
            z_roots[z_root_index] = c_div(c_add(x, d), a2);
            z_roots[z_root_index + 1] = c_div(c_sub(x, d), a2);
            z_root_index += 2;
        }
        if (m == 1) {
            z_roots[z_root_index] = c_div(c_neg(coeffs[0]), coeffs[1]);
            z_root_index += 1;
        }
        if (z_root_index + 1 > len) status = VBXO_ERR_PANIC;
        else {
            for (size_t i = 0; i < z_root_index + 1; i++) self[i] = z_roots[i];
            for (size_t i = z_root_index + 1; i < len; i++) self[i] = c_new(0, 0);
        }
    }
    free(z_roots); free(rem); free(coeffs);
    return status;
}

/* polynomial.rs:79-89 */
int vbxo_find_roots_f32(const vbxo_c32 *p, size_t len, vbxo_c32 *roots, size_t *n_roots) {
    vbxo_c32 *other = (vbxo_c32 *)malloc(len * sizeof(vbxo_c32));
    memcpy(other, p, len * sizeof(vbxo_c32));
    int st = vbxo_find_roots_mut_f32(other, len);
    size_t l = len;
    if (st == VBXO_OK) {
        while (l > 0 && c_is_zero(other[l - 1])) l--;
        memcpy(roots, other, l * sizeof(vbxo_c32));
        *n_roots = l;
    } else *n_roots = 0;
    free(other);
    return st;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Sample = f32 instantiation of the slice traits: the same statements as vbx_oracle.c with every
 * value and every intermediate held in `float`, which is what the generic code monomorphised at T = f32 computes
 * (src/periodic.rs:276-289 add_amp / mul_amp on f32; src/waves.rs:60-76; src/spectrum.rs:62-146; :401-441).
 * The build uses -ffp-contract=off and no fast-math, so each operation rounds to f32 once, as in Rust.
 * Parity unpinned by the reference: none of its tests instantiates these traits at f32.
 * ------------------------------------------------------------------------------------------------------------------ */
void vbxo_autocorrelate_f32(const float *x, size_t n, float *coeffs, size_t n_lags) {        /* periodic.rs:280-287 */
    for (size_t lag = 0; lag < n_lags; lag++) {
        float accum = x[0];
        for (size_t i = 1; i + lag < n; i++) accum = accum + x[i] * x[i + lag];
        coeffs[lag] = accum;
    }
}

void vbxo_normalize_f32(float *x, size_t n) {                                                  /* waves.rs:44-75 */
    float m = (x[0] < 0.0f) ? x[0] * -1.0f : x[0];
    for (size_t i = 1; i < n; i++) {
        float a = (x[i] < 0.0f) ? x[i] * -1.0f : x[i];
        if (a > m) m = a;
    }
    float scale = 1.0f / m;
    for (size_t i = 0; i < n; i++) x[i] = x[i] * scale;
}

void vbxo_lpc_f32(const float *r, size_t n_coeffs, float *ac, float *kc_out) {                 /* spectrum.rs:62-92 */
    float kc[64], tmp[64];
    for (size_t i = 0; i < n_coeffs + 1; i++) ac[i] = 0.0f;
    for (size_t i = 0; i < n_coeffs; i++) { kc[i] = 0.0f; tmp[i] = 0.0f; }
    float err = r[0];
    ac[0] = 1.0f;
    for (size_t i = 1; i < n_coeffs + 1; i++) {
        float acc = r[i];
        for (size_t j = 1; j < i; j++) acc = acc + (ac[j] * r[i - j]);
        kc[i - 1] = -acc / err;
        ac[i] = kc[i - 1];
        for (size_t j = 0; j < n_coeffs; j++) tmp[j] = ac[j];
        for (size_t j = 1; j < i; j++) ac[j] = ac[j] + (kc[i - 1] * tmp[i - j]);
        err = err * (1.0f - (kc[i - 1] * kc[i - 1]));
    }
    if (kc_out) for (size_t i = 0; i < n_coeffs; i++) kc_out[i] = kc[i];
}

int vbxo_lpc_burg_f32(const float *x, size_t n, size_t n_coeffs, float *coeffs) {              /* spectrum.rs:101-146 */
    if (n < 2) return 4;
    float *b1 = (float *)calloc(n, sizeof(float));
    float *b2 = (float *)calloc(n, sizeof(float));
    float *aa = (float *)calloc(n_coeffs ? n_coeffs : 1, sizeof(float));
    int status = 0;
    b1[0] = x[0];
    b2[n - 2] = x[n - 1];
    for (size_t j = 2; j < n; j++) { b1[j - 1] = x[j - 1]; b2[j - 2] = x[j - 1]; }
    for (size_t i = 1; i < n_coeffs + 1; i++) {
        float num = 0.0f, denum = 0.0f;
        for (size_t j = 1; j + i < n + 1; j++) {
            num = num + b1[j - 1] * b2[j - 1];
            denum = denum + b1[j - 1] * b1[j - 1] + b2[j - 1] * b2[j - 1];     /* powi(2) == x * x */
        }
        if (denum <= 0.0f) { status = 1; break; }
        coeffs[i - 1] = 2.0f * num / denum;
        for (size_t j = 1; j < i; j++) coeffs[j - 1] = aa[j - 1] - coeffs[i - 1] * aa[i - j - 1];
        if (i < n_coeffs) {
            for (size_t j = 1; j < i + 1; j++) aa[j - 1] = coeffs[j - 1];
            for (size_t j = 1; j + i < n; j++) {
                b1[j - 1] = b1[j - 1] - aa[i - 1] * b2[j - 1];
                b2[j - 1] = b2[j] - aa[i - 1] * b1[j];
            }
        }
    }
    if (status == 0) for (size_t c = 0; c < n_coeffs; c++) coeffs[c] = coeffs[c] * -1.0f;
    free(b1); free(b2); free(aa);
    return status;
}

/* MFCC at T = f32 (spectrum.rs:410-441): the transform runs on Complex<f32> (rustfft, un-vendored: the mathematical DFT is
 * evaluated in double and each bin rounded to f32, i.e. an ideally rounded f32 transform); norm_sqr / norm in f32; the
 * filter sums, the log and the clamp in f64 (`to_f64`, `0f64` fold); the energies are rounded to f32 (`T::from_f64`) and
 * dct() on &[f32] widens each energy, accumulates in f64 and rounds the doubled sum to f32 once. */
void vbxo_mfcc_bins(size_t n, size_t num_coeffs, double lo, double hi, double sample_rate, size_t *bins);
int vbxo_mfcc_f32(const float *x, size_t n, size_t num_coeffs, double lo, double hi, double sample_rate, float *out) {
    size_t *bins = (size_t *)malloc((num_coeffs + 2) * sizeof(size_t));
    vbxo_mfcc_bins(n, num_coeffs, lo, hi, sample_rate, bins);
    size_t top = bins[num_coeffs + 1];
    if (top > n) { free(bins); return 4; }
    float *sre = (float *)calloc(n, sizeof(float)), *sim = (float *)calloc(n, sizeof(float));
    double *ct = (double *)malloc(n * sizeof(double)), *st = (double *)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; i++) { double ang = 2.0 * M_PI * (double)i / (double)n; ct[i] = cos(ang); st[i] = sin(ang); }
    for (size_t k = bins[0]; k < top; k++) {
        double re = 0., im = 0.;
        size_t idx = 0;
        for (size_t i = 0; i < n; i++) { re += (double)x[i] * ct[idx]; im -= (double)x[i] * st[idx]; idx += k; if (idx >= n) idx -= n; }
        sre[k] = (float)re; sim[k] = (float)im;
    }
    free(ct); free(st);
    float *energies = (float *)malloc(num_coeffs * sizeof(float));
    for (size_t wdx = 0; wdx < num_coeffs; wdx++) {
        size_t w0 = bins[wdx], w1 = bins[wdx + 1], w2 = bins[wdx + 2];
        size_t up = w1 - w0, down = w2 - w1;
        double up_sum = 0., down_sum = 0.;
        for (size_t i = 0, bin = w0; bin < w1; i++, bin++) {
            float ns = sre[bin] * sre[bin] + sim[bin] * sim[bin];                    /* Complex<f32>::norm_sqr */
            up_sum = up_sum + fabs((double)ns) * ((double)i / (double)up);
        }
        for (size_t i = 0, bin = w1; bin < w2; i++, bin++) {
            float nm = hypotf(sre[bin], sim[bin]);                                   /* Complex<f32>::norm */
            down_sum = down_sum + fabs((double)nm) * ((double)i / (double)down);
        }
        double lg = log10(up_sum + down_sum);
        energies[wdx] = (float)((isnan(lg) || lg < 1.0e-10) ? 1.0e-10 : lg);
    }
    for (size_t k = 0; k < num_coeffs; k++) {                                        /* dct_mut on &[f32], spectrum.rs:391-398 */
        double acc = 0.;                                                             /* fold(0., ..): f64 accumulator */
        for (size_t i = 0; i < num_coeffs; i++)
            acc = acc + (double)energies[i] * cos(M_PI * (double)k * (2. * (double)i + 1.) / (2. * (double)num_coeffs));
        out[k] = (float)(2. * acc);                                                  /* T::from_f64 */
    }
    free(bins); free(sre); free(sim); free(energies);
    return 0;
}
