// vbx_roots.hpp -- the reference's root finder, operation by operation (src/polynomial.rs:26-152), and Resonance::from_root
// (src/spectrum.rs:166-192), shared by k_roots.hip (the Polynomial trait's entry points, the reference-faithful
// find_formants kernel) and k_roots_fast.hip (which runs it on the frames its own method hands on).
#pragma once

#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

constexpr int ROOTS_BLOCK = 64;


template <typename T>
struct lds_poly_t {
    cx<T> *base;   // element j of this lane at base[j * ROOTS_BLOCK]
    __device__ __forceinline__ cx<T> get(int j) const { return base[j * ROOTS_BLOCK]; }
    __device__ __forceinline__ void set(int j, cx<T> v) const { base[j * ROOTS_BLOCK] = v; }
};
using lds_poly = lds_poly_t<double>;

// src/polynomial.rs:26-32
template <typename T>
__device__ __forceinline__ int poly_degree(const lds_poly_t<T> &p, int len) {
    int d = 0;
    for (int j = 0; j < len; j++) if (!ciszero(p.get(j))) d = j;
    return d;
}
template <typename T>
__device__ __forceinline__ int poly_off_low(const lds_poly_t<T> &p, int len) {
    int d = -1;
    for (int j = len - 1; j >= 0; j--) if (!ciszero(p.get(j))) d = j;
    return d < 0 ? 0 : d;
}

// src/polynomial.rs:34-72.  n = len - 1 stays fixed across deflations (Q11).
// top (<= n): every coefficient above index `top` is zero (deflation clears them).  The reference's Horner chains run
// through those zeros -- 0 * z + 0 three times per index, which leaves all three accumulators exactly zero for finite
// z -- so starting at `top` is the same arithmetic with the no-ops left out (37 % of the Horner steps of an order-12
// polynomial).  A non-finite z (0 * inf = NaN in the reference) takes the full chain.
template <typename T>
__device__ __forceinline__ cx<T> laguerre(const lds_poly_t<T> &p, int len, cx<T> start, int top = -1) {
    const int n = len - 1;
    const T dn = (T)n, dnn1 = (T)(n - 1) * (T)n;
    cx<T> z = start;
    bool done = false;
    if (top < 0 || top > n) top = n;
    // the highest degree among the ACTIVE lanes, as a scalar: entries above a lane's own degree are zeros, and a chain
    // that starts above it runs through 0 * z + 0 (see above).  The callers reach this point diverged (find_roots_emit's
    // early returns and its per-lane trip count), so the maximum is built bit by bit from ballots, which only ever see
    // the active lanes -- a shuffle butterfly would read inactive lanes' registers.
    int top_u = 0;
    for (int bit = 64; bit > 0; bit >>= 1)                   // top <= n <= 64 (len <= 65 by the API's bound)
        if (__any(top >= (top_u | bit))) top_u |= bit;
    for (int it = 0; it < 20; it++) {
        const bool zfin = (z.re - z.re == T(0)) && (z.im - z.im == T(0));
        cx<T> a0, a1 = cmk<T>(T(0), T(0)), a2 = cmk<T>(T(0), T(0));
        if (__all(zfin || done)) {
            // the usual case: the chain starts at the wave's highest degree, a scalar -- the loop runs on scalar control
            // (unrolled by two: the three accumulators rotate through the same registers without copies)
            const int hi = top_u;
            a0 = p.get(hi);
#pragma unroll 2
            for (int j = hi - 1; j >= 0; j--) {
                a2 = cmad(a2, z, a1);
                a1 = cmad(a1, z, a0);
                a0 = cmad(a0, z, p.get(j));
            }
        } else {
            a0 = p.get(n);                                   // a non-finite z somewhere: every lane's full chain
            for (int j = n - 1; j >= 0; j--) {
                a2 = cmad(a2, z, a1);
                a1 = cmad(a1, z, a0);
                a0 = cmad(a0, z, p.get(j));
            }
        }
        // |p(z)| <= 1e-16  (compared on squared norms)
        const T n0 = a0.re * a0.re + a0.im * a0.im;
        if (!done && n0 <= T(1.0e-32)) done = true;
        if (__all(done)) break;
        const cx<T> ca = cdiv(cneg(a1), a0);
        const cx<T> ca2 = cmul(ca, ca);
        const cx<T> cb = csub(ca2, cdiv(cmk<T>(T(2) * a2.re, T(2) * a2.im), a0));
        const cx<T> c1 = csqrt(csub(cmk<T>(dnn1 * cb.re, dnn1 * cb.im), ca2));
        const cx<T> cc1 = cadd(ca, c1), cc2 = csub(ca, c1);
        const T m1 = cc1.re * cc1.re + cc1.im * cc1.im, m2 = cc2.re * cc2.re + cc2.im * cc2.im;
        const cx<T> den = (m1 > m2) ? cc1 : cc2;
        const cx<T> cc = cdiv(cmk<T>(dn, T(0)), den);
        if (!done) z = cadd(z, cc);
    }
    return z;
}

// src/polynomial.rs:92-152 (+ div_polynomial_mut :155-195 inlined as in-place synthetic division).
// co: polynomial in / scratch; emit(index, root) receives the roots in discovery order.
template <typename T, typename Emit>
__device__ __forceinline__ int find_roots_emit(const lds_poly_t<T> &co, int len, Emit emit) {
    const int coeff_high = poly_degree(co, len);
    if (coeff_high < 1) return 2;                       // Err(Polynomial), :95
    const int coeff_low = poly_off_low(co, len);
    if (coeff_low > 0) return 4;                        // coeffs[co] out of bounds, :110-112
    int m = coeff_high;
    const int clen = coeff_high + 1;
    int zi = 0;
    for (int k = m; k >= 3; k--) {                      // (3..m+1).rev()
        // the lanes of a wave hold different polynomials: the chain starts at the highest degree among them
        const int ns = poly_degree(co, clen);          // this lane's degree (the division below starts from it)
        const cx<T> z = laguerre(co, clen, cmk<T>(T(-2), T(-2)), ns);   // laguerre() takes the wave's maximum of it
        emit(zi++, z);
        if (ciszero(z)) return 2;                       // div by zero -> Err, :123,:192
        // divide by (x - z): other = -z; q[i] = c[i+1] - q[i+1]*other
        cx<T> t = co.get(ns);
        for (int i = ns - 1; i >= 0; i--) {
            const cx<T> old = co.get(i);
            co.set(i, t);
            t = cmad(t, z, old);                        // old - t*(-z)
        }
        co.set(ns, cmk<T>(T(0), T(0)));
        m -= 1;
    }
    if (m == 2) {                                       // :131-139
        const cx<T> c0 = co.get(0), c1 = co.get(1), c2 = co.get(2);
        const cx<T> a2 = cadd(c2, c2);
        const cx<T> four_ac = cmul(cmk<T>(T(4) * c2.re, T(4) * c2.im), c0);
        const cx<T> d = csqrt(csub(cmul(c1, c1), four_ac));
        const cx<T> xx = cneg(c1);
        emit(zi, cdiv(cadd(xx, d), a2));
        emit(zi + 1, cdiv(csub(xx, d), a2));
        zi += 2;
    } else if (m == 1) {                                // :141-144
        emit(zi, cdiv(cneg(co.get(0)), co.get(1)));
        zi += 1;
    }
    return 0;
}

// src/spectrum.rs:166-192
__device__ __forceinline__ bool resonance_from_root(c64 root, double sample_rate, res_t &out) {
    const double freq_mul = sample_rate / (M_PI * 2.0);
    if (!(root.im >= 0.0)) return false;
    double r = hypot(root.re, root.im), theta = atan2(root.im, root.re);
    if (r > 1.0) {      // root.conj().inv() = (re, im) / |root|^2
        const double ns = root.re * root.re + root.im * root.im;
        const double ire = root.re / ns, iim = root.im / ns;
        r = hypot(ire, iim); theta = atan2(iim, ire);
    }
    const double frequency = freq_mul * theta;
    const double bandwidth = -2.0 * freq_mul * log(r);
    if (frequency > 50.0 && frequency < sample_rate * 0.5 - 50.0) {
        out.frequency = frequency; out.bandwidth = bandwidth;
        return true;
    }
    return false;
}

// insertion of one resonance into a frequency-sorted row (stable: equal keys keep arrival order)
__device__ __forceinline__ void res_insert_sorted(res_t *row, int count, res_t v) {
    int j = count;
    while (j > 0 && row[j - 1].frequency > v.frequency) { row[j] = row[j - 1]; j--; }
    row[j] = v;
}

}  // namespace vbx
