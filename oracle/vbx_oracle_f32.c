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
