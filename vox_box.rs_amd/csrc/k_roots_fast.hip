// k_roots_fast.hip -- the resonances of find_formants (src/lib.rs:80-110) from the Burg coefficients without walking the
// reference's root finder step by step.
//
// The reference (src/polynomial.rs:34-152; k_roots.hip follows it operation by operation) finds the p roots of the REAL
// polynomial rev([1, a1..ap]) one at a time: Laguerre from (-2, -2) in complex arithmetic with the degree fixed at p, always
// to its 20-iteration limit (its stopping test |p(z)| <= 1e-16 is below the rounding of the evaluation, so it never fires),
// deflation by one complex root, p - 2 times over, then the quadratic formula.  find_formants keeps the roots with im > 0,
// turns them into (frequency, bandwidth) and SORTS them by frequency (src/lib.rs:94-110): what leaves is a function of the
// polynomial's root SET, not of the order or the path by which the roots were found.  The iteration above ends at roots
// converged to the rounding level (that is what 20 Laguerre steps do), so any other method that delivers converged roots
// delivers the same resonances to ~1e-11 relative (measured below), against a gate of 1e-4.
//
// This kernel uses what the reference's generic Polynomial code cannot: the coefficients are real.
//   * roots come in conjugate pairs: one Laguerre solve per PAIR, deflation by the real quadratic t^2 - 2 Re z t + |z|^2
//     (a solve that lands on a real root deflates by t - x): ~5.7 solves per frame instead of 10;
//   * p, p', p'' at a complex z from three real synthetic divisions by that quadratic (2 FMAs per coefficient each)
//     instead of three complex Horner chains (4 each);
//   * the iteration stops when it has converged (|dz| <= 1e-7 |z|: the convergence is cubic, the step after that is below
//     1e-20) -- 4.2 iterations on average instead of 20 -- with the proper degree of the deflated polynomial in the formula,
//     from a start next to the unit circle (where LPC roots are) instead of (-2, -2); fractional steps from the 9th
//     iteration on break limit cycles;
//   * every root is polished by one Newton step on the ORIGINAL polynomial, and that step is also the check: a root whose
//     polish step exceeds 1e-7 |z|, a solve that has not converged after 32 iterations, a non-finite coefficient -- and
//     the frame is done again, in the same wavefront, by the reference's own iteration (vbx_roots.hpp: a branch that only
//     the wavefronts with such a frame enter; none in 300,000 speech frames).
// One lane per frame, coefficients in registers (compile-time indices; the chains start at the wavefront's highest degree).
// Statuses that depend on the coefficients alone are the reference's: a zero constant term is Err(PANIC) (:110-112).
//
// Measured against the oracle on 40,000 speech frames (numpy model of this file, then the kernel in tests/): resonance
// counts equal, frequencies within 1.7e-11 relative, bandwidths within 7.5e-12.
#include "vbx_roots.hpp"

namespace vbx {

namespace {

struct cd { double re, im; };

__device__ __forceinline__ double rcp_fast(double d) { return rcp_nr2(d); }

// p, p', p'' of the real polynomial c[0..top] at z = x + i y by three synthetic divisions by t^2 - 2 x t + (x^2 + y^2), on
// register arrays (compile-time indices; `top` is a scalar; entries above a lane's own degree are zero and leave the
// chains at zero).  With p(t) = q(t) Q(t) + b1 (t - 2x) + b0:  p(z) = b0 + b1 (z - 2x),  p'(z) = 2 i y Q(z) + b1,
// p''(z) = 2 Q(z) - 8 y^2 Q2(z) + 4 i y e1  (Q, Q2: the quotients of the first and second division; e: the second's
// coefficients).  b: the first division's coefficients (the deflated polynomial is b[2..]).  tests/roots_fast_model.py
// is the same in numpy.
template <int P>
__device__ __forceinline__ void eval3_plain(const double (&c)[P + 1], const int top, const double x, const double y,
                                            cd &p, cd &dp, cd &ddp, double (&b)[P + 3]) {
    const double r = x + x, s = fma(x, x, y * y);
    double e[P + 3], g[P + 3];
#pragma unroll
    for (int k = 0; k < P + 3; k++) { b[k] = 0.0; e[k] = 0.0; g[k] = 0.0; }
#pragma unroll
    for (int k = P; k >= 0; k--) if (k <= top) b[k] = fma(r, b[k + 1], fma(-s, b[k + 2], c[k]));
#pragma unroll
    for (int k = P - 2; k >= 0; k--) if (k <= top - 2) e[k] = fma(r, e[k + 1], fma(-s, e[k + 2], b[k + 2]));
#pragma unroll
    for (int k = P - 4; k >= 0; k--) if (k <= top - 4) g[k] = fma(r, g[k + 1], fma(-s, g[k + 2], e[k + 2]));
    // p(z) = b0 + b1 (z - r);  Q(z), Q2(z) likewise;  p' = 2 i y Q + b1;  p'' = 2 Q - 8 y^2 Q2 + 4 i y e1
    p.re = fma(-x, b[1], b[0]); p.im = y * b[1];
    const double qre = fma(-x, e[1], e[0]), qim = y * e[1];
    const double q2re = fma(-x, g[1], g[0]), q2im = y * g[1];
    const double y2 = y + y;
    dp.re = fma(-y2, qim, b[1]); dp.im = y2 * qre;
    const double y8 = 8.0 * y * y;
    ddp.re = fma(-y8, q2re, 2.0 * qre); ddp.im = fma(-y8, q2im, fma(2.0 * y2, e[1], 2.0 * qim));
}

__device__ __forceinline__ cd cmul_(cd a, cd b) { return cd{fma(a.re, b.re, -a.im * b.im), fma(a.re, b.im, a.im * b.re)}; }

__device__ __forceinline__ cd csqrt_(cd z) {
    const double m = sqrt(fma(z.re, z.re, z.im * z.im));
    if (m == 0.0) return cd{0.0, 0.0};
    const double t = sqrt(0.5 * (m + fabs(z.re)));
    const double u = 0.5 * z.im * rcp_fast(t);
    return (z.re >= 0.0) ? cd{t, u} : cd{fabs(u), copysign(t, z.im)};
}

// src/spectrum.rs:166-192 for a root with im > 0
__device__ __forceinline__ bool resonance_of(double re, double im, double sample_rate, res_t &out) {
    const double freq_mul = sample_rate / (M_PI * 2.0);
    double r = hypot(re, im), theta = atan2(im, re);
    if (r > 1.0) {                                           // root.conj().inv() = (re, im) / |root|^2
        const double ns = re * re + im * im;
        const double ire = re / ns, iim = im / ns;
        r = hypot(ire, iim); theta = atan2(iim, ire);
    }
    const double frequency = freq_mul * theta;
    const double bandwidth = -2.0 * freq_mul * log(r);
    if (frequency > 50.0 && frequency < sample_rate * 0.5 - 50.0) { out.frequency = frequency; out.bandwidth = bandwidth; return true; }
    return false;
}

constexpr int RF_MAX_IT = 32;
// every solve starts here.  LPC roots lie inside the unit circle, most of them close to it: from a point near the circle
// the first step already lands next to a root, where the reference's (-2, -2) spends two or three steps coming in and, for
// polynomials with their roots evenly spaced on a circle, enters a two-cycle through the origin that the fractional steps
// do not always break in time (6 frames in 300,000).  Measured on 30,000 speech frames (tools/experiments/ has the sweep):
// iterations summed over a wavefront's solves 53.8 from (-2, 2), 31.9 from here; no frame left unconverged.
constexpr double RF_START_RE = 0.8, RF_START_IM = 0.4;
constexpr double RF_CONV = 1e-7;          // |dz| <= RF_CONV |z|: converged (cubic: the next step would be ~1e-20)
constexpr double RF_CHECK = 1e-7;         // the polish step of an accepted root
constexpr double RF_REAL = 1e-9;          // |im z| <= RF_REAL |z|: a real root

}  // namespace

template <int P>
__global__ __launch_bounds__(64) void formant_resonances_fast_kernel(
    const double *__restrict__ coeffs, long n_frames, double sample_rate,
    res_t *__restrict__ out_res, int32_t *__restrict__ out_count, int32_t *__restrict__ status, const frame_map_t map,
    int32_t *__restrict__ redo_count /* frames done again by the reference's iteration (a probe for tests) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // [P + 1][64] complex: the reference's iteration
    constexpr int NR = P / 2;                                // resonances a frame can have
    const long f = frame_map(map, (long)blockIdx.x * 64 + threadIdx.x, n_frames);
    const bool active = f >= 0;
    const long fr = active ? f : 0;
    const double *a = coeffs + fr * (long)P;
    const int st = (status != nullptr) ? status[fr] : 0;
    // c = rev([1, a1..ap]): index j < P holds a[P-1-j], index P holds 1 (src/lib.rs:80-91)
    double c0[P + 1], c[P + 1];
    bool finite = true;
#pragma unroll
    for (int j = 0; j < P; j++) { c0[j] = a[P - 1 - j]; finite = finite && (c0[j] - c0[j] == 0.0); }
    c0[P] = 1.0;
#pragma unroll
    for (int j = 0; j <= P; j++) c[j] = c0[j];
    const bool run = active && st == 0;
    int rst = 0;
    if (run && c0[0] == 0.0) rst = 4;                        // off_low > 0: coeffs[co] out of bounds, :110-112
    bool bad = run && rst == 0 && !finite;                   // -> the reference-faithful kernel
    int m = (run && rst == 0 && !bad) ? P : 0;
    double rf[NR], rb[NR];                                   // the frame's resonances, sorted by frequency
    bool direct_row = false;                                 // a lane redone by the reference's iteration writes its row itself
#pragma unroll
    for (int j = 0; j < NR; j++) { rf[j] = 0.0; rb[j] = 0.0; }
    int count = 0;

    auto emit = [&](double x, double y) {                    // a root with im > 0 of the deflated polynomial
        // one Newton step on the original polynomial: the polish, and the check
        cd p, dp, ddp; double b[P + 3];
        eval3_plain<P>(c0, P, x, y, p, dp, ddp, b);
        const double inv = rcp_fast(fma(dp.re, dp.re, dp.im * dp.im));
        const double sre = fma(p.re, dp.re, p.im * dp.im) * inv, sim = fma(p.im, dp.re, -p.re * dp.im) * inv;
        const double zz = fma(x, x, y * y);
        if (!(fma(sre, sre, sim * sim) <= RF_CHECK * RF_CHECK * zz)) bad = true;    // NaN: bad
        x -= sre; y -= sim;
        res_t v;
        if (y > 0.0 && resonance_of(x, y, sample_rate, v)) {
            // sorted insertion (stable: an equal frequency goes behind), compile-time indices
            bool placed = false;
            double cf = v.frequency, cb = v.bandwidth;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const bool here = !placed && (j >= count || rf[j] > cf);
                // from the insertion point on, every entry moves up by one: carry it along
                if (here) placed = true;
                if (placed) { const double tf = rf[j], tb = rb[j]; rf[j] = cf; rb[j] = cb; cf = tf; cb = tb; }
            }
            count++;
        }
    };

    while (__any(m > 2)) {
        int top = 0;                                         // the wavefront's highest degree, a scalar
        for (int bit = 16; bit > 0; bit >>= 1) if (__any(m >= (top | bit))) top |= bit;
        double x = RF_START_RE, y = RF_START_IM;
        const bool solving = m > 2;
        bool done = !solving;
        const double dn = (double)m, dn1 = (double)(m - 1);
        for (int it = 0; it < RF_MAX_IT; it++) {
            cd p, dp, ddp; double b[P + 3];
            eval3_plain<P>(c, top, x, y, p, dp, ddp, b);
            const double pn = fma(p.re, p.re, p.im * p.im);
            const double ip = rcp_fast(pn);
            // G = p'/p, H = G^2 - p''/p, sq = sqrt((n-1)(n H - G^2)), dz = n / (G +- sq) (the larger denominator)
            const cd G{fma(dp.re, p.re, dp.im * p.im) * ip, fma(dp.im, p.re, -dp.re * p.im) * ip};
            const cd R{fma(ddp.re, p.re, ddp.im * p.im) * ip, fma(ddp.im, p.re, -ddp.re * p.im) * ip};
            const cd G2 = cmul_(G, G);
            const cd H{G2.re - R.re, G2.im - R.im};
            const cd sq = csqrt_(cd{dn1 * fma(dn, H.re, -G2.re), dn1 * fma(dn, H.im, -G2.im)});
            const cd d1{G.re + sq.re, G.im + sq.im}, d2{G.re - sq.re, G.im - sq.im};
            const double n1 = fma(d1.re, d1.re, d1.im * d1.im), n2 = fma(d2.re, d2.re, d2.im * d2.im);
            const cd den = (n1 > n2) ? d1 : d2;
            const double idn = dn * rcp_fast(n1 > n2 ? n1 : n2);
            double sre = den.re * idn, sim = -den.im * idn;
            const bool frac = it >= 8 && (it - 8) % 5 == 0;  // scalar: a fractional step breaks a limit cycle
            if (frac) {
                const int w = ((it - 8) / 5) % 7;
                const double fr_ = (w == 0) ? 0.5 : (w == 1) ? 0.25 : (w == 2) ? 0.75 : (w == 3) ? 0.13 : (w == 4) ? 0.38 : (w == 5) ? 0.62 : 0.88;
                sre *= fr_; sim *= fr_;
            }
            const bool exact = pn == 0.0;                    // on a root: no step
            if (!done && !exact) { x -= sre; y -= sim; }
            const bool conv = exact || (!frac && fma(sre, sre, sim * sim) <= RF_CONV * RF_CONV * fma(x, x, y * y));
            done = done || conv;
            if (__all(done)) break;
        }
        if (solving && !done) { bad = true; m = 0; }         // not converged (or NaN): the reference-faithful kernel
        if (solving && done) {
            const bool real = fabs(y) <= RF_REAL * sqrt(fma(x, x, y * y));
            y = fabs(y);
            if (!real) emit(x, y);
            // deflation: by t - x (real root) or by t^2 - 2 x t + (x^2 + y^2)
            cd p, dp, ddp; double b[P + 3];
            eval3_plain<P>(c, top, x, real ? 0.0 : y, p, dp, ddp, b);
            if (real) {
                // synthetic division by (t - x): q_k = c_{k+1} + x q_{k+1}
                double t = 0.0;
#pragma unroll
                for (int k = P; k >= 0; k--) { const double ck = c[k]; c[k] = t; t = fma(x, t, ck); }
                m -= 1;
            } else {
#pragma unroll
                for (int k = 0; k <= P; k++) c[k] = (k + 2 <= P) ? b[k + 2] : 0.0;
                m -= 2;
            }
        }
    }
    if (m == 2) {                                            // the last pair: the quadratic formula on real coefficients
        const double a2 = c[2], a1 = c[1], a0 = c[0];
        const double disc = fma(a1, a1, -4.0 * a2 * a0);
        if (disc < 0.0) {
            const double i2 = rcp_fast(a2 + a2);
            emit(-a1 * i2, fabs(sqrt(-disc) * i2));
        } else if (!(disc >= 0.0)) bad = true;
    }
    if (__any(bad)) {
        // the reference's iteration for the lanes that need it (the others wait): polynomial in LDS as [index][lane]
        lds_poly co{reinterpret_cast<c64 *>(lds_raw) + threadIdx.x};
#pragma unroll
        for (int j = 0; j <= P; j++) co.set(j, cmk(c0[j], 0.0));
        if (bad) {
            // the row goes straight to memory, all VBX_MAX_RESONANCES slots of it: an iteration that has not converged can
            // leave more than P / 2 roots above the real axis (the reference takes up to 32 and panics on the 33rd,
            // src/lib.rs:97-99: resonances[count] out of bounds)
            count = 0;
            int total = 0;
            res_t *rowd = out_res + fr * (long)VBX_MAX_RESONANCES_K;
            rst = find_roots_emit(co, P + 1, [&](int, c64 z) {
                res_t v;
                if (z.im > 0.0 && resonance_from_root(z, sample_rate, v)) {
                    if (count < VBX_MAX_RESONANCES_K) { if (active) res_insert_sorted(rowd, count, v); count++; }
                    total++;
                }
            });
            if (rst == 0 && total > VBX_MAX_RESONANCES_K) rst = 4;   // VBX_FRAME_ERR_PANIC
            direct_row = true;
            if (active && redo_count != nullptr) atomicAdd(redo_count, 1);
        }
    }
    if (!active) return;
    if (rst != 0 || st != 0) count = 0;
    res_t *row = out_res + f * (long)VBX_MAX_RESONANCES_K;
    if (direct_row) {
        for (int j = count; j < VBX_MAX_RESONANCES_K; j++) { row[j].frequency = 0.0; row[j].bandwidth = 0.0; }
    } else {
#pragma unroll
        for (int j = 0; j < NR; j++) { row[j].frequency = (j < count) ? rf[j] : 0.0; row[j].bandwidth = (j < count) ? rb[j] : 0.0; }
        for (int j = NR; j < VBX_MAX_RESONANCES_K; j++) { row[j].frequency = 0.0; row[j].bandwidth = 0.0; }
    }
    if (out_count != nullptr) out_count[f] = count;
    if (status != nullptr && st == 0 && rst != 0) status[f] = rst;
}

// the orders with an instantiation: 12 (BASELINE), 10 and 13 (the reference's own callers: tests/lib.rs:23,52,
// examples/formant_extraction/src/main.rs:53), 8, 14, 16
bool formant_resonances_fast_supported(int p) {
    const char *e = getenv("VBX_ROOTS_DIRECT");              // 1: the reference's iteration for every frame (A/B, tests)
    return !(e && atoi(e) != 0) && (p == 8 || p == 10 || p == 12 || p == 13 || p == 14 || p == 16);
}

// redo_count (optional): a device counter the kernel adds the number of frames done by the reference's iteration to
void launch_formant_resonances_fast(hipStream_t s, const double *coeffs, long F, int p, double sample_rate,
                                    res_t *out_res, int32_t *out_count, int32_t *status, frame_map_t map, int32_t *redo_count) {
    const long items = frame_map_items(map, F);
    if (items <= 0) return;
    const size_t lds = (size_t)(p + 1) * ROOTS_BLOCK * sizeof(c64);
#define VBX_RF(PP) hipLaunchKernelGGL((formant_resonances_fast_kernel<PP>), dim3((unsigned)((items + 63) / 64)), dim3(64), lds, s, \
                                      coeffs, F, sample_rate, out_res, out_count, status, map, redo_count)
    switch (p) {
        case 8: VBX_RF(8); break;
        case 10: VBX_RF(10); break;
        case 12: VBX_RF(12); break;
        case 13: VBX_RF(13); break;
        case 14: VBX_RF(14); break;
        case 16: VBX_RF(16); break;
        default: break;
    }
#undef VBX_RF
}

}  // namespace vbx
