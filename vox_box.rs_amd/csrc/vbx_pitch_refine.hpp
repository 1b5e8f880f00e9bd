// vbx_pitch_refine.hpp -- the part of Pitched::pitch that follows the autocorrelation: peak scan, exact top-k
// bounds, Brent/sinc refinement and the sorted candidate list.  Shared by the two kernels that produce the lag curve
// y in LDS: pitch_kernel (k_pitch.hip: all-lag autocorrelation on the FP64 matrix cores, any frame length) and
// analyze_kernel (k_spectral.hip: autocorrelation, LPC and MFCC from one FFT of the frame).
//
// Reference: src/periodic.rs:29-87 (interpolate_sinc), :103-188 (brent_maximize), :192-229 (improve_extremum),
//            :362-375 (local_maxima), :413-455 (pitch, after the lag window division).  Quirks Q4-Q10 reproduced.
#pragma once
#include <type_traits>

#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {


// p * x + c with the CONSTANT c in a scalar register pair.  Left to the compiler, a Horner step with a 64-bit literal becomes
// v_mov_b32 x2 (the literal into the accumulator) + v_fmac_f64: three vector instructions per coefficient, two of them moves --
// and these kernels are bound by vector-instruction issue (the four polynomials of one sinc evaluation were 125 of its ~350
// instructions, 83 of them such moves).  As v_fma_f64 with a scalar addend the literal is built by the scalar unit.
__device__ __forceinline__ double fma_sc(double p, double x, const double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(x), "s"(c));
    return r;
}

// sin(x), cos(x) for |x| <= pi/2 (a little beyond is fine): Taylor to x^21 / x^22 (< 2e-18 truncation)
__device__ __forceinline__ double sin_poly(double x) {
    const double x2 = x * x;
    double p = -1.9572941063391261231e-20;           // -1/21!
    p = fma_sc(p, x2, 8.2206352466243297170e-18);    //  1/19!
    p = fma_sc(p, x2, -2.8114572543455207632e-15);   // -1/17!
    p = fma_sc(p, x2, 7.6471637318198164759e-13);    //  1/15!
    p = fma_sc(p, x2, -1.6059043836821614599e-10);   // -1/13!
    p = fma_sc(p, x2, 2.5052108385441718775e-08);    //  1/11!
    p = fma_sc(p, x2, -2.7557319223985890653e-06);   // -1/9!
    p = fma_sc(p, x2, 1.9841269841269841270e-04);    //  1/7!
    p = fma_sc(p, x2, -8.3333333333333333333e-03);   // -1/5!
    p = fma_sc(p, x2, 1.6666666666666666667e-01);    //  1/3!
    return x * fma(-x2, p, 1.0);
}
__device__ __forceinline__ double cos_poly(double x) {
    const double x2 = x * x;
    double p = 8.8967913924505732867e-22;            //  1/22!
    p = fma_sc(p, x2, -4.1103176233121648585e-19);   // -1/20!
    p = fma_sc(p, x2, 1.5619206968586226462e-16);    //  1/18!
    p = fma_sc(p, x2, -4.7794773323873852974e-14);   // -1/16!
    p = fma_sc(p, x2, 1.1470745597729724714e-11);    //  1/14!
    p = fma_sc(p, x2, -2.0876756987868098979e-09);   // -1/12!
    p = fma_sc(p, x2, 2.7557319223985890653e-07);    //  1/10!
    p = fma_sc(p, x2, -2.4801587301587301587e-05);   // -1/8!
    p = fma_sc(p, x2, 1.3888888888888888889e-03);    //  1/6!
    p = fma_sc(p, x2, -4.1666666666666666667e-02);   // -1/4!
    p = fma(p, x2, 0.5);                             //  1/2!
    return fma(-x2, p, 1.0);
}

// cos(theta) for theta in [0, pi]: -sin(theta - pi/2)
__device__ __forceinline__ double cos_0_pi(double theta) { return -sin_poly(theta - 1.57079632679489661923); }

// y lookup: entries [nvalid, ylen) are the zeros of self_lag.resize(2N, 0) (src/periodic.rs:411)
__device__ __forceinline__ double y_at(const double *y, int nvalid, int idx) {
    return (idx < nvalid) ? y[idx] : 0.0;
}
// the same without a branch where the curve is followed by stored zeros (pitch_refine_store's LDS image: Y_PAD of them
// after y[n), nvalid = n + Y_PAD): an index past them reads the last of those zeros (never the array that follows the image)
__device__ __forceinline__ double y_at_padded(const double *y, int nvalid, int idx) {
    return y[(idx < nvalid) ? idx : nvalid - 1];
}


// ------------------------------------------------------------------------------------------
// lane groups: G consecutive lanes cooperate on one candidate / query point (group_sum<G> in
// vbx_device.hpp)
// ------------------------------------------------------------------------------------------
// General form of the sinc sum (src/periodic.rs:59-84): any index may need the reference's
// clamps.  Even lanes of a group take "left" terms, odd lanes "right" terms.  No cross-lane ops.
template <int G>
__device__ __forceinline__ double sinc_terms_general(const double *y, int nvalid, int ylen, int offset, int nl, int nr,
                                                     double phil, double phir, int max_depth, int lig = -1) {
    if (lig < 0) lig = lane_id() & (G - 1);    // (the two-lanes-in-one form passes the lane it stands in for)
    const int side = lig & 1;
    const double ph = side ? phir : phil;
    const double s0 = sinpi(ph);               // sin(pi*(ph+n)) = (-1)^n * s0
    const double inv_dd = 1.0 / (ph + (double)max_depth);
    const int ibase = side ? (offset + nl) : (offset + nr);
    double acc = 0.0;
    for (int n = (lig >> 1); n <= max_depth; n += G / 2) {
        const double a = M_PI * (ph + (double)n);
        int idx = side ? (ibase + n) : (ibase - n);
        idx = (idx < 0) ? 0 : idx;
        idx = (idx >= ylen) ? (ylen - 1) : idx;          // only reachable on the right side (:78)
        const double r_lag = y_at(y, nvalid, idx);
        const double sgn_s0 = (n & 1) ? -s0 : s0;
        const double first = sgn_s0 * rcp_nr2(a);
        const double second = fma(0.5, cos_0_pi(a * inv_dd), 0.5);
        acc = fma(r_lag * first, second, acc);
    }
    return acc;
}

// The same terms when every index is known to be in [0, nvalid): no clamps, and the per-term work
// is reduced with exact identities (no change of the reference's formula):
//   sin(pi*(ph+n))          = (-1)^n * sin(pi*ph)                       one polynomial per evaluation
//   0.5 + 0.5*cos(a/(ph+D)) = 0.5 + 0.5*cos(theta0 + j*delta)           a lane's terms (n = n0 + j*G/2) are
//                             equally spaced in angle: Reinsch's stable cosine recurrence
//                             d += -kappa*C; C += d  (kappa = 4 sin^2(delta/2)) replaces the cosine
//   1/(ph+n)                = four terms share ONE v_rcp_f64 (+ a Newton step): 1/(p0 p1 p2 p3), then products
// Returns the lane's partial sum already scaled (sum over the group = interpolate_sinc).
// One lane's share of that sum as a stream: the terms n = n0, n0 + NSTEP, ... of one side (left: indices descend from
// ibase_l - n0, right: ascend from ibase_r + n0).
struct sinc_stream_t {
    double pn, C, d, kappa, acc0, acc1;                        // ph + n; cos(theta_j), C_j - C_{j-1}; sum t, sum t*C  (t = y/(ph+n))
    const double *yp;
};

template <int NSTEP>
__device__ __forceinline__ void sinc_stream_init(sinc_stream_t &s, const double *y, int side, int n0, int ibase_l, int ibase_r,
                                                 double phil, double phir, int max_depth) {
    constexpr double S = (double)NSTEP;
    const double ph = side ? phir : phil;
    const double h2 = M_PI * rcp_nr2(ph + (double)max_depth); // theta = h2 * (ph + n)  in [0, pi]
    s.pn = ph + (double)n0;
    // cosine recurrence state: C = cos(theta_j), d = C_j - C_{j-1}
    const double theta0 = h2 * s.pn, delta = h2 * S;
    s.C = cos_0_pi(theta0);
    s.d = cos_0_pi(theta0 + delta) - s.C;                      // only used when the lane has >= 2 terms
    const double sh = sin_poly(0.5 * delta);
    s.kappa = 4.0 * sh * sh;
    s.yp = y + (side ? (ibase_r + n0) : (ibase_l - n0));
    s.acc0 = 0.0; s.acc1 = 0.0;
}

// four terms; step = +-NSTEP (the side's direction)
template <int NSTEP>
__device__ __forceinline__ void sinc_stream_block4(sinc_stream_t &s, const int step) {
    constexpr double S = (double)NSTEP;
    const double y0 = s.yp[0], y1 = s.yp[step], y2 = s.yp[2 * step], y3 = s.yp[3 * step];
    const double p0 = s.pn, p1 = s.pn + S, p2 = s.pn + 2.0 * S, p3 = s.pn + 3.0 * S;
    const double q01 = p0 * p1, q23 = p2 * p3;
    const double r = rcp_nr1(q01 * q23);
    const double r01 = r * q23, r23 = r * q01;
    const double t0 = y0 * (r01 * p1), t1 = y1 * (r01 * p0), t2 = y2 * (r23 * p3), t3 = y3 * (r23 * p2);
    s.acc0 += t0; s.acc1 = fma(t0, s.C, s.acc1);
    s.C += s.d; s.d = fma(-s.kappa, s.C, s.d);                 // after the first step d_1 was preset: see above
    s.acc0 += t1; s.acc1 = fma(t1, s.C, s.acc1);
    s.C += s.d; s.d = fma(-s.kappa, s.C, s.d);
    s.acc0 += t2; s.acc1 = fma(t2, s.C, s.acc1);
    s.C += s.d; s.d = fma(-s.kappa, s.C, s.d);
    s.acc0 += t3; s.acc1 = fma(t3, s.C, s.acc1);
    s.C += s.d; s.d = fma(-s.kappa, s.C, s.d);
    s.pn += 4.0 * S;
    s.yp += 4 * step;
}

// 1..3 terms left: one more block, padded with zero-weight terms
template <int NSTEP>
__device__ __forceinline__ void sinc_stream_tail(sinc_stream_t &s, const int step, const int rem) {
    constexpr double S = (double)NSTEP;
    const double y0 = s.yp[0], y1 = (rem > 1) ? s.yp[step] : 0.0, y2 = (rem > 2) ? s.yp[2 * step] : 0.0;
    const double p0 = s.pn, p1 = s.pn + S, p2 = s.pn + 2.0 * S;
    const double q01 = p0 * p1;
    const double r = rcp_nr1(q01 * p2);
    const double r01 = r * p2;
    const double t0 = y0 * (r01 * p1), t1 = y1 * (r01 * p0), t2 = y2 * (r * q01);
    s.acc0 += t0; s.acc1 = fma(t0, s.C, s.acc1);
    s.C += s.d; s.d = fma(-s.kappa, s.C, s.d);
    s.acc0 += t1; s.acc1 = fma(t1, s.C, s.acc1);
    s.C += s.d;
    s.acc0 += t2; s.acc1 = fma(t2, s.C, s.acc1);
}

// the lane's partial sum, scaled and ROUNDED before the cross-lane sum (vbx_device.hpp); k = sin(pi*ph) / pi * 0.5 (the taper's)
__device__ __forceinline__ double sinc_stream_result(const sinc_stream_t &s, int n0, double k) {
    const double acc = s.acc0 + s.acc1;
    return rounded(((n0 & 1) ? -acc : acc) * k);
}

template <int G>
__device__ __forceinline__ double sinc_terms_fast(const double *y, int ibase_l, int ibase_r,
                                                  double phil, double phir, int max_depth) {
    constexpr int NSTEP = G / 2;
    const int lig = lane_id() & (G - 1);
    const int side = lig & 1;
    const int n0 = lig >> 1;
    const double s0 = sin_poly(M_PI * fmin(phil, phir));      // sin(pi*phil) == sin(pi*phir)
    const int nterms = (n0 <= max_depth) ? (max_depth - n0) / NSTEP + 1 : 0;
    sinc_stream_t s;
    sinc_stream_init<NSTEP>(s, y, side, n0, ibase_l, ibase_r, phil, phir, max_depth);
    const int step = side ? NSTEP : -NSTEP;
    int j = 0;
    for (; j + 4 <= nterms; j += 4) sinc_stream_block4<NSTEP>(s, step);
    if (j < nterms) sinc_stream_tail<NSTEP>(s, step, nterms - j);
    return sinc_stream_result(s, n0, s0 * (0.5 * 0.31830988618379067154));
}

// sinc_terms_fast<64> with what depends on the abscissa's unit cell alone handed in (improve_extremum_sinc_wave keeps it
// across the evaluations of a Brent run): yp0 = the lane's first sample, nterms its term count, dmd = (double)max_depth.
// Every floating-point operation is sinc_terms_fast<64>'s, in its order.
__device__ __forceinline__ double sinc_terms_fast_cell(const double *yp0, int side, int n0, int nterms, double dmd,
                                                       double phil, double phir) {
    constexpr int NSTEP = 32;
    constexpr double S = (double)NSTEP;
    const int step = side ? NSTEP : -NSTEP;
    const double s0 = sin_poly(M_PI * fmin(phil, phir));      // sin(pi*phil) == sin(pi*phir)
    sinc_stream_t s;
    const double ph = side ? phir : phil;
    const double h2 = M_PI * rcp_nr2(ph + dmd);               // theta = h2 * (ph + n)  in [0, pi]
    s.pn = ph + (double)n0;
    const double theta0 = h2 * s.pn, delta = h2 * S;
    s.C = cos_0_pi(theta0);
    s.d = cos_0_pi(theta0 + delta) - s.C;
    const double sh = sin_poly(0.5 * delta);
    s.kappa = 4.0 * sh * sh;
    s.yp = yp0;
    s.acc0 = 0.0; s.acc1 = 0.0;
    int j = 0;
    for (; j + 4 <= nterms; j += 4) sinc_stream_block4<NSTEP>(s, step);
    if (j < nterms) sinc_stream_tail<NSTEP>(s, step, nterms - j);
    return sinc_stream_result(s, n0, s0 * (0.5 * 0.31830988618379067154));
}

// The same sum with each lane standing in for TWO lanes of a group of VG: lane p of a group of VG / 2 runs the left stream
// (lane 2p of the VG) and the right stream (lane 2p + 1) of n0 = p side by side and adds the two results -- the first level of
// group_sum<VG>'s tree, whose other levels are group_sum<VG / 2> over the half-sized group (quad_xor1, quad_rev,
// row_half_mirror there pair the same partial sums as quad_rev, row_half_mirror, row_mirror here; every level's addition
// is commutative).  Same operations on the same values: group_sum<VG / 2> of this is bit for bit group_sum<VG> of
// sinc_terms_fast<VG>.  What it buys: the per-evaluation work that is NOT the terms (the Brent bookkeeping, the early-outs)
// is shared by twice as many candidates per wavefront.
template <int VG>
__device__ __forceinline__ double sinc_terms_fast_dual(const double *y, int ibase_l, int ibase_r,
                                                       double phil, double phir, int max_depth) {
    constexpr int NSTEP = VG / 2;
    const int n0 = lane_id() & (VG / 2 - 1);
    const double s0 = sin_poly(M_PI * fmin(phil, phir));
    const int nterms = (n0 <= max_depth) ? (max_depth - n0) / NSTEP + 1 : 0;
    sinc_stream_t sl, sr;
    sinc_stream_init<NSTEP>(sl, y, 0, n0, ibase_l, ibase_r, phil, phir, max_depth);
    sinc_stream_init<NSTEP>(sr, y, 1, n0, ibase_l, ibase_r, phil, phir, max_depth);
    int j = 0;
    for (; j + 4 <= nterms; j += 4) { sinc_stream_block4<NSTEP>(sl, -NSTEP); sinc_stream_block4<NSTEP>(sr, NSTEP); }
    if (j < nterms) { sinc_stream_tail<NSTEP>(sl, -NSTEP, nterms - j); sinc_stream_tail<NSTEP>(sr, NSTEP, nterms - j); }
    const double k = s0 * (0.5 * 0.31830988618379067154);
    return sinc_stream_result(sl, n0, k) + sinc_stream_result(sr, n0, k);
}

// interpolate_sinc (src/periodic.rs:29-87), cooperative over groups of G lanes; the arguments are
// uniform inside a group and the result is bit-identical in all its lanes.  Lanes with
// active == false contribute nothing.  st |= 4 where the reference would index out of bounds.
// y[0..nvalid) is readable (entries past the data are zero), ylen >= nvalid is the logical length
// after self_lag.resize(2N, 0).  Must be called from converged code.
// trusted: sinc_bracket_trusted() has shown that EVERY abscissa of the caller's bracket takes the clamp-free sum or one of
// the two exact-integer early-outs with readable indices; the range tests of :38-40 and the index checks are then skipped
// (they cannot fire), the arithmetic is the same.
// DUAL: G lanes stand in for a group of 2G (sinc_terms_fast_dual): the value is bit for bit sinc_interp<2G>'s.
template <int G, bool DUAL = false>
__device__ __forceinline__ double sinc_interp(const double *y, int nvalid, int ylen, int offset, int nx,
                                              double x, int max_depth, bool active, int &st,
                                              unsigned *terms = nullptr, bool trusted = false) {
    bool summed = false, fast = false;
    double special = 0.0, phil = 0.5, phir = 0.5;
    int nl = 0, nr = 1;
    if (active && trusted) {
        const double fl = floor(x);
        nl = (int)fl;
        nr = nl + 1;
        phil = x - fl;
        phir = 1.0 - phil;
        if (fabs(x - (double)nl) < 1.0e-10) special = y[offset + nl];                          // :41
        else if (fabs(x - (double)nr) < 1.0e-10) special = y[offset + nr];                     // :42
        else {
            if ((offset + nr) < max_depth) max_depth = offset + nr;                            // :46-52 (offset + nr >= 1 here)
            if ((offset + nl + max_depth) >= nx) max_depth = nx - offset + nl - 1;             // :55-57
            summed = true; fast = true;
        }
    } else if (active) {
        if (nx < 1) special = __builtin_nan("");                                   // :38
        else if (x > (double)nx) {                                                 // :39
            const int idx = offset + nx - 1;
            if (idx < 0 || idx >= ylen) st |= 4; else special = y_at(y, nvalid, idx);
        } else if (x < 0.0) special = y_at(y, nvalid, 0);                          // :40
        else {
            const double fl = floor(x);
            nl = (fl > 0.0) ? (int)fl : 0;                                          // NaN -> 0
            nr = nl + 1;
            phil = x - (double)nl;
            phir = 1.0 - phil;
            if (fabs(x - (double)nl) < 1.0e-10) {                                   // :41
                const int idx = offset + nl;
                if (idx < 0 || idx >= ylen) st |= 4; else special = y_at(y, nvalid, idx);
            } else if (fabs(x - (double)nr) < 1.0e-10) {                            // :42
                const int idx = offset + nr;
                if (idx < 0 || idx >= ylen) st |= 4; else special = y_at(y, nvalid, idx);
            } else {
                if ((offset + nr) < max_depth) max_depth = ((offset + nr) < 0) ? 0 : (offset + nr);   // :46-52
                if ((offset + nl + max_depth) >= nx) max_depth = nx - offset + nl - 1;                  // :55-57
                if (max_depth < 0 || offset + nr >= ylen) st |= 4;   // usize wrap / left index at n = 0 out of bounds
                else {
                    summed = true;
                    // all left indices [offset+nr-D, offset+nr] and right indices [offset+nl, offset+nl+D] readable
                    fast = !(x != x) && max_depth <= offset + nr && offset + nl >= 0 &&
                           offset + nl + max_depth < nvalid && offset + nr < nvalid;
                }
            }
        }
    }
    double acc = 0.0;
    if (terms != nullptr && summed) *terms += 2u * (unsigned)(max_depth + 1);
    if (summed) {
        if constexpr (DUAL) {
            if (fast) acc = sinc_terms_fast_dual<2 * G>(y, offset + nr, offset + nl, phil, phir, max_depth);
            else {
                const int p = lane_id() & (G - 1);
                acc = rounded(sinc_terms_general<2 * G>(y, nvalid, ylen, offset, nl, nr, phil, phir, max_depth, 2 * p)) +
                      rounded(sinc_terms_general<2 * G>(y, nvalid, ylen, offset, nl, nr, phil, phir, max_depth, 2 * p + 1));
            }
        } else {
            if (fast) acc = sinc_terms_fast<G>(y, offset + nr, offset + nl, phil, phir, max_depth);
            else acc = sinc_terms_general<G>(y, nvalid, ylen, offset, nl, nr, phil, phir, max_depth);
        }
    }
    const double total = group_sum<G>(acc);
    return summed ? total : special;
}

// True when every abscissa in [a0, b0] is handled by sinc_interp's clamp-free sum or by an exact-integer early-out whose
// index is readable: the same conditions sinc_interp tests per evaluation, tested once for each of the (at most three)
// unit cells the bracket touches.  False for NaN bounds.
__device__ __forceinline__ bool sinc_bracket_trusted(double a0, double b0, int nvalid, int ylen, int offset, int nx, int depth) {
    if (!(a0 >= 0.0 && b0 <= (double)nx && b0 - a0 <= 2.5)) return false;
    const int lo = (int)floor(a0), hi = (int)floor(b0);
    bool ok = true;
    for (int nl = lo; nl <= hi; nl++) {
        const int nr = nl + 1;
        int D = depth;
        if ((offset + nr) < D) D = ((offset + nr) < 0) ? 0 : (offset + nr);
        if ((offset + nl + D) >= nx) D = nx - offset + nl - 1;
        ok = ok && D >= 0 && offset + nr < ylen && D <= offset + nr && offset + nl >= 0 && offset + nl + D < nvalid &&
             offset + nr < nvalid && offset + nr >= 1;
    }
    return ok;
}

// improve_extremum(.., Interpolation::Sinc(depth), true) (src/periodic.rs:192-229) around
// brent_maximize (:103-188): a MINIMISER of the un-negated interpolant (Q8).  One candidate per
// group of G lanes; `active` lanes carry a candidate.  Must be called from converged code.
template <int G>
__device__ __forceinline__ void improve_extremum_sinc(const double *y, int nvalid, int ylen, int offset, int nx,
                                                      double ixmid, int depth, bool active,
                                                      double &xmid, double &ymid, int &st,
                                                      unsigned *terms = nullptr, unsigned *evals = nullptr,
                                                      double bar = -__builtin_inf(), bool *pruned = nullptr,
                                                      bool negate = false /* is_max == false: the closure negates, :219-222 */) {
#pragma clang fp contract(off)   // keep the scalar iteration bit-identical to the unfused CPU arithmetic
    const double golden = 1. - 0.6180339887498948482045868343656381177203091798057628621;
    const double sqrt_epsilon = 1.4901161193847656e-08;   // sqrt(f64::EPSILON)
    const double eps = 2.220446049250313e-16;
    const double tol = 1e-10;
    xmid = 0.; ymid = 0.;
    bool run = active;
    if (active) {
        if (ixmid == 0.) { xmid = 0.; ymid = y_at(y, nvalid, 0); run = false; }                 // :193
        else if (ixmid >= (double)nx) {                                                          // :194
            run = false;
            if (nx < 1 || nx - 1 >= ylen) st |= 4;
            else { xmid = (double)nx; ymid = y_at(y, nvalid, nx - 1); }
        } else if (!(ixmid - 1. < ixmid + 1.)) { st |= 4; run = false; }                        // assert!(a < b), :113
    }
    double a = ixmid - 1., b = ixmid + 1.;
    const bool trusted = run && sinc_bracket_trusted(a, b, nvalid, ylen, offset, nx, depth);
    double v = a + golden * (b - a);
    double fv = sinc_interp<G>(y, nvalid, ylen, offset, nx, v, depth, run, st, terms, trusted);
    if (negate) fv = -fv;
    if (evals != nullptr && run) *evals += 1u;
    double x = v, w = v, fx = fv, fw = fv;
    bool done = !run;
    // a >= -offset: every abscissa of the bracket has its left neighbour at index >= 0, so none of the evaluations a
    // pruned candidate skips could have been an out-of-bounds panic of the reference
    const bool prunable = pruned != nullptr && run && a >= (double)(-offset);
    if (pruned != nullptr) {     // exact top-k pruning (pitch_refine_kernel): f(v0) already caps the final strength
        const double ub = (fv <= 1.) ? fv : 1.;
        *pruned = prunable && ub < bar;
        done = done || *pruned;
    }
    for (int it = 1; it <= 60; it++) {
        const double range = b - a;
        const double middle_range = (a + b) * 0.5;
        const double tol_act = sqrt_epsilon * fabs(x) + tol / 3.;
        if (!done && fabs(x - middle_range) + range * 0.5 <= 2. * tol_act) done = true;
        if (!__any(!done)) break;
        double new_step = (x < middle_range) ? golden * (b - x) : golden * (a - x);
        if (fabs(x - w) >= tol_act) {
            const double t = (x - w) * (fx - fv);
            double q = (x - v) * (fx - fw);
            double p = (x - v) * q - (x - w) * t;
            q = 2. * q - t;
            if (q > 0.) p = -p; else q = -q;
            if (fabs(p) < fabs(new_step * q) && p > q * (a - x + 2. * tol_act) && p < q * (b - x - 2. * tol_act))
                new_step = p / q;
        }
        if (fabs(new_step) < tol_act) new_step = (new_step > 0.) ? tol_act : -tol_act;
        const double t = x + new_step;
        double ft = sinc_interp<G>(y, nvalid, ylen, offset, nx, t, depth, !done, st, terms, trusted);
        if (negate) ft = -ft;
        if (evals != nullptr && !done) *evals += 1u;
        if (!done) {
            if (ft <= fx) {
                if (t < x) b = x; else a = x;
                v = w; w = x; x = t;
                fv = fw; fw = fx; fx = ft;
            } else {
                if (t < x) a = t; else b = t;
                if (ft <= fw || fabs(w - x) < eps) {
                    v = w; w = t;
                    fv = fw; fw = ft;
                } else if (ft <= fv || fabs(v - x) < eps || fabs(v - w) < eps) {
                    v = t;
                    fv = ft;
                }
            }
            // brent_maximize only ever replaces fx by a smaller value (:162): the final strength is <= the current fx
            // (<= 1 here, so the reflection of :446 does not apply).  Strictly below the bar it cannot be returned.
            if (prunable && fx < bar) { *pruned = true; done = true; }
        }
    }
    if (run) { xmid = x; ymid = fx; }
}

// The same refinement when the whole wavefront works on ONE candidate (G = 64, every lane active, every argument identical
// in all lanes) and the bracket is trusted (sinc_bracket_trusted): every value of the Brent iteration is then identical
// in all lanes, so every decision is taken ONCE, as a scalar branch (__any of a uniform condition is the condition), instead
// of being if-converted into per-lane selects whose both sides execute: the parabolic step with its IEEE division runs only
// when it is taken (a quarter of the steps), the bookkeeping of (a, b, v, w, x, fv, fw, fx) becomes register moves on the
// taken path.  Same operations on the same values in the same order as improve_extremum_sinc<64>: bit-identical results
// (tests/test_gpu_parity.py::test_improve_extremum_points compares the two forms).  Returns false when the bracket is not
// trusted or an early-out applies (the caller then takes the general form).
__device__ __forceinline__ bool improve_extremum_sinc_wave(const double *y, int nvalid, int ylen, int offset, int nx,
                                                           double ixmid, int depth, double &xmid, double &ymid,
                                                           unsigned &terms, unsigned &evals, double bar, bool &pruned) {
#pragma clang fp contract(off)   // keep the scalar iteration bit-identical to the unfused CPU arithmetic
    const double golden = 1. - 0.6180339887498948482045868343656381177203091798057628621;
    const double sqrt_epsilon = 1.4901161193847656e-08;   // sqrt(f64::EPSILON)
    const double eps = 2.220446049250313e-16;
    const double tol = 1e-10;
    if (__any(ixmid == 0. || ixmid >= (double)nx || !(ixmid - 1. < ixmid + 1.))) return false;   // :193-194, :113
    double a = ixmid - 1., b = ixmid + 1.;
    if (!__any(sinc_bracket_trusted(a, b, nvalid, ylen, offset, nx, depth))) return false;
    // one evaluation of the interpolant at a trusted abscissa (sinc_interp<64>'s trusted arm, decisions as scalar branches).
    // Round 5, same operations on the same values with fewer instructions around them:
    //  * the two exact-integer tests of :41-42 read phil / phir: x - (double)nl IS phil (x - floor(x) is exact), and
    //    (double)nr - x is 1 - phil exactly wherever it could be below 1e-10 (phil >= 0.5: Sterbenz);
    //  * everything that depends on the unit CELL of x alone -- the clamped depth, each lane's term count and first sample's
    //    address, the terms counter's increment -- is kept from the previous evaluation: a Brent run converges inside one or
    //    two cells (the bracket is two lags wide), so these ~18 integer instructions run once or twice per candidate
    //    instead of once per evaluation.
    const int lane = lane_id();
    const int side = lane & 1, n0 = lane >> 1;
    // two cells are kept: the iteration closes in on an integer lag -- a jump of the mirrored interpolant (Q6) -- from BOTH
    // sides, so its abscissae alternate between the two cells next to it
    struct cell_t { int nl, nterms; unsigned tinc; double dmd; const double *yp0; };
    cell_t ca{-0x7fffffff, 0, 0u, 0.0, y}, cb{-0x7fffffff, 0, 0u, 0.0, y};
    bool a_is_older = true;
    auto fill = [&](cell_t &c, int nl) {
        c.nl = nl;
        const int nr = nl + 1;
        int md = depth;
        if ((offset + nr) < md) md = offset + nr;                                                // :46-52 (offset + nr >= 1 here)
        if ((offset + nl + md) >= nx) md = nx - offset + nl - 1;                                 // :55-57
        c.tinc = 2u * (unsigned)(md + 1);
        c.dmd = (double)md;
        c.nterms = (n0 <= md) ? (md - n0) / 32 + 1 : 0;
        c.yp0 = y + (side ? (offset + nl + n0) : (offset + nr - n0));
    };
    auto eval = [&](double x) -> double {
        const double fl = floor(x);
        const int nl = (int)fl;
        const double phil = x - fl, phir = 1.0 - phil;
        if (__any(phil < 1.0e-10)) return y[offset + nl];                                        // :41
        if (__any(phir < 1.0e-10)) return y[offset + nl + 1];                                    // :42
        if (__any(nl == ca.nl)) {
            a_is_older = false;
            terms += ca.tinc;
            return group_sum<64>(sinc_terms_fast_cell(ca.yp0, side, n0, ca.nterms, ca.dmd, phil, phir));
        }
        if (__any(nl != cb.nl)) {
            if (a_is_older) { fill(ca, nl); a_is_older = false; terms += ca.tinc;
                              return group_sum<64>(sinc_terms_fast_cell(ca.yp0, side, n0, ca.nterms, ca.dmd, phil, phir)); }
            fill(cb, nl);
        }
        a_is_older = true;
        terms += cb.tinc;
        return group_sum<64>(sinc_terms_fast_cell(cb.yp0, side, n0, cb.nterms, cb.dmd, phil, phir));
    };
    double v = a + golden * (b - a);
    double fv = eval(v);
    evals += 1u;
    double x = v, w = v, fx = fv, fw = fv;
    // a >= -offset: every abscissa of the bracket has its left neighbour at index >= 0, so none of the evaluations a
    // pruned candidate skips could have been an out-of-bounds panic of the reference
    const bool prunable = __any(a >= (double)(-offset));
    pruned = false;
    if (prunable && __any(((fv <= 1.) ? fv : 1.) < bar)) { pruned = true; return true; }          // f(v0) already caps the strength
    for (int it = 1; it <= 60; it++) {
        const double range = b - a;
        const double middle_range = (a + b) * 0.5;
        const double tol_act = sqrt_epsilon * fabs(x) + tol / 3.;
        if (__any(fabs(x - middle_range) + range * 0.5 <= 2. * tol_act)) break;
        double new_step = __any(x < middle_range) ? golden * (b - x) : golden * (a - x);
        if (__any(fabs(x - w) >= tol_act)) {
            const double t = (x - w) * (fx - fv);
            double q = (x - v) * (fx - fw);
            double p = (x - v) * q - (x - w) * t;
            q = 2. * q - t;
            if (__any(q > 0.)) p = -p; else q = -q;
            if (__any(fabs(p) < fabs(new_step * q) && p > q * (a - x + 2. * tol_act) && p < q * (b - x - 2. * tol_act))) {
                asm volatile("" ::: "memory");   // a real branch: the IEEE division runs when the step is taken (a quarter of
                new_step = p / q;                // the iterations), not speculated on every one and then selected
            }
        }
        if (__any(fabs(new_step) < tol_act)) new_step = __any(new_step > 0.) ? tol_act : -tol_act;
        const double t = x + new_step;
        const double ft = eval(t);
        evals += 1u;
        if (__any(ft <= fx)) {
            if (__any(t < x)) b = x; else a = x;
            v = w; w = x; x = t;
            fv = fw; fw = fx; fx = ft;
        } else {
            if (__any(t < x)) a = t; else b = t;
            if (__any(ft <= fw || fabs(w - x) < eps)) {
                v = w; w = t;
                fv = fw; fw = ft;
            } else if (__any(ft <= fv || fabs(v - x) < eps || fabs(v - w) < eps)) {
                v = t;
                fv = ft;
            }
        }
        // brent_maximize only ever replaces fx by a smaller value (:162): the final strength is <= the current fx
        // (<= 1 here, so the reflection of :446 does not apply).  Strictly below the bar it cannot be returned.
        if (prunable && __any(fx < bar)) { pruned = true; return true; }
    }
    xmid = x; ymid = fx;
    return true;
}

// ------------------------------------------------------------------------------------------
// Pitched::pitch after the lag curve (src/periodic.rs:413-455), one wavefront per frame:
//  a) peak scan, lane-parallel: strict local maxima of y[0..N/2) (Q4), the "parabolic" lag (Q5) and
//     the frequency filter (:439); survivors are compacted in index order into an LDS list.
//     The sinc(30) strength of :433 is dead in the reference (overwritten at :448 for every
//     candidate that passes the filter, dropped otherwise) and is not evaluated.
//  b, c) refinement (improve_extremum_sinc) and the sorted candidate list, see pitch_refine_store.
// ------------------------------------------------------------------------------------------
constexpr int Y_PAD = 16;                       // zeros kept after y[n) (stand for the head of resize(2N, 0))
#ifndef VBX_EXP_PB
#define VBX_EXP_PB 5
#endif
constexpr int PB = VBX_EXP_PB;                  // lags per block of the |y| prefix sums (<= Y_PAD + 1: the last block reads on into the zeros)
// sum of |y| over one block (the same association wherever it is taken)
__device__ __forceinline__ double block_abs_sum(const double *yp) {
    double s = fabs(yp[0]) + fabs(yp[1]);
#pragma unroll
    for (int i = 2; i + 1 < PB; i += 2) s += fabs(yp[i]) + fabs(yp[i + 1]);
    if (PB & 1) s += fabs(yp[PB - 1]);
    return s;
}
typedef unsigned short cand_t;                  // candidate lags (< 2048)
constexpr int PG = 16;                          // lanes per query point (sinc_points / extremum_points kernels)
constexpr int PNG = 64 / PG;                    // points per wavefront
constexpr int GROUP_PATH_MIN_CAND = 32;         // pitch frames with more candidates refine them 4 at a time, 16 lanes each
#ifndef VBX_EXP_RG
#define VBX_EXP_RG 16
#endif
#ifndef VBX_EXP_GROUP_KMAX
#define VBX_EXP_GROUP_KMAX 4                    // kmax from which frames with few candidates take the group path too
#endif
constexpr int RG = VBX_EXP_RG;                  // lanes per candidate on the group path of pitch_refine_store
#ifndef VBX_EXP_DUAL_MIN_CAND
#define VBX_EXP_DUAL_MIN_CAND 6
#endif
constexpr int DUAL_MIN_CAND = VBX_EXP_DUAL_MIN_CAND;   // unpruned frames with at least this many candidates: 8 groups, two lanes in one

__device__ __forceinline__ void cand_from_peak(const double *ys, int kk, double sample_rate, int offset,
                                               double &freq, double &nn, const bool f32 = false) {
    const double peak = ys[kk], peak_rev = ys[kk - 1], peak_fwd = ys[kk + 1];
    if (f32) {                                                        // T = S::Float = f32: differences, quotients in f32
#pragma clang fp contract(off)
        const double dr = 0.5 * (double)(float)(peak_fwd - peak_rev);
        const double d2r = 2. * peak - (double)(float)(peak_rev - peak_fwd);
        freq = (double)(float)(sample_rate / (double)(float)((double)kk + dr / d2r));
        nn = (double)(float)((double)(float)(sample_rate / freq) - (double)offset);
        return;
    }
    const double dr = 0.5 * (peak_fwd - peak_rev);                    // :423
    const double d2r = 2. * peak - (peak_rev - peak_fwd);             // :424 (Q5)
    freq = sample_rate / ((double)kk + dr / d2r);                     // :425
    nn = sample_rate / freq - (double)offset;                         // :432, :443
}

constexpr int BOUND_HEAD = 8;                   // nearest terms per side evaluated by the first-evaluation bound

// 0.5 + 0.5 cos(theta) from ABOVE for theta in [0, pi]: the cosine series cut after + theta^8 / 8! (from there on its terms
// decrease, so the remainder is <= 0); within 1.3e-2 of the taper at pi, 1e-4 at pi / 2, exact to 1e-9 below 0.5.  The tail
// of the first-evaluation bound only needs an upper bound of the taper at the start of each range, and it needs it cheaply:
// noise-like frames spend half their time here (175 candidates, none refined).
__device__ __forceinline__ double taper_upper(double theta) {
    const double t = theta * theta;
    double p = fma(t, 2.48015873015873016e-05, -1.38888888888888894e-03);   // 1/8!, -1/6!
    p = fma(p, t, 4.16666666666666644e-02);                                  // 1/4!
    p = fma(p, t, -0.5);
    return fma(0.5, fma(p, t, 1.0), 0.5) * (1.0 + 1.0e-15);
}

// sum of |y_i| over i in [i0, i1], rounded outward to blocks of PB (p16[j] = sum_{i < PB j} |y_i|, j <= nblk)
__device__ __forceinline__ double abs_range_bound(const double *p16, int nblk, int i0, int i1) {
    i0 = (i0 < 0) ? 0 : i0;
    const int last = PB * nblk - 1;
    i1 = (i1 > last) ? last : i1;
    if (i1 < i0) return 0.0;
    return p16[i1 / PB + 1] - p16[i0 / PB];
}

// Upper bound of f(v0), the FIRST value brent_maximize (src/periodic.rs:103-188) takes on the bracket
// [nn-1, nn+1]: v0 = a + golden*(b-a).  Every later accepted value is <= f(v0), so this bounds the candidate's
// final strength from above.  One lane per candidate: the 2*BOUND_HEAD terms nearest to v0 are summed, the rest
// is bounded by  sum |y_i| * c(n)  with  c(n) = |sin(pi ph)| * taper(n) / (pi (ph + n)),  decreasing in n, taken
// at the start of geometrically growing ranges.  +inf (= "refine it") for everything that is not the plain
// clamp-free sum, and for NaN/inf data.
__device__ __forceinline__ double first_eval_bound(const double *ys, const double *p16, int nblk, int nvalid, int ylen,
                                                   int offset, int nx, double nn, int depth) {
    const double INF = __builtin_inf();
    if (nn == 0. || nn >= (double)nx || !(nn - 1. < nn + 1.)) return INF;
    double v0;
    {
#pragma clang fp contract(off)
        const double golden = 1. - 0.6180339887498948482045868343656381177203091798057628621;
        const double ba = nn - 1., bb = nn + 1.;
        if (!(ba >= (double)(-offset))) return INF;
        v0 = ba + golden * (bb - ba);
    }
    if (nx < 1 || v0 > (double)nx || v0 < 0.0) return INF;
    const double fl = floor(v0);
    const int nl = (int)fl, nr = nl + 1;
    const double phil = v0 - (double)nl, phir = 1.0 - phil;
    if (fabs(v0 - (double)nl) < 1.0e-10 || fabs(v0 - (double)nr) < 1.0e-10) return INF;
    int D = depth;
    if ((offset + nr) < D) D = ((offset + nr) < 0) ? 0 : (offset + nr);            // :46-52
    if ((offset + nl + D) >= nx) D = nx - offset + nl - 1;                          // :55-57
    if (D < 0 || offset + nr >= ylen || D > offset + nr || offset + nl < 0 || offset + nl + D >= ylen) return INF;
    const double s0 = sin_poly(M_PI * fmin(phil, phir)) * 0.31830988618379067154;   // |sin(pi ph)| / pi >= 0
    const double hl = M_PI * rcp_nr1(phil + (double)D), hr = M_PI * rcp_nr1(phir + (double)D);
    double head = 0.0;
    const int nh = (BOUND_HEAD < D + 1) ? BOUND_HEAD : D + 1;
    {   // the nh nearest terms per side, exactly: their taper angles hl (phil + m), hr (phir + m) are equally spaced, so two
        // cosines per side start Reinsch's recurrence (C += d; d -= kappa C, kappa = 4 sin^2(step / 2)) instead of nh
        double cl = cos_0_pi(hl * phil), cr = cos_0_pi(hr * phir);
        double dl = cos_0_pi(hl * (phil + 1.0)) - cl, dr = cos_0_pi(hr * (phir + 1.0)) - cr;
        const double sl = sin_poly(0.5 * hl), sr = sin_poly(0.5 * hr);
        const double kl = 4.0 * sl * sl, kr = 4.0 * sr * sr;
        // the 2 nh samples first, all in flight together: read where they are used, every term waited a full LDS latency
        // for its own sample (16 dependent waits per candidate pass; same operations on the same values)
        double yl[BOUND_HEAD], yr[BOUND_HEAD];
#pragma unroll
        for (int m = 0; m < BOUND_HEAD; m++) {
            const bool in = m < nh;
            yl[m] = y_at_padded(ys, nvalid, in ? offset + nr - m : 0);
            yr[m] = y_at_padded(ys, nvalid, in ? offset + nl + m : 0);
        }
#pragma unroll
        for (int m = 0; m < BOUND_HEAD; m++) {
            if (m >= nh) break;
            const double pl = phil + (double)m, pr = phir + (double)m;
            const double tl = yl[m] * rcp_nr1(pl) * fma(0.5, cl, 0.5);
            const double tr = yr[m] * rcp_nr1(pr) * fma(0.5, cr, 0.5);
            const double t = tl + tr;
            head += (m & 1) ? -t : t;
            cl += dl; dl = fma(-kl, cl, dl);
            cr += dr; dr = fma(-kr, cr, dr);
        }
    }
    head *= s0;
    double tail = 0.0;
    for (int lo = nh; lo <= D; lo *= 2) {
        const int hi = (2 * lo < D + 1) ? 2 * lo : D + 1;                            // terms [lo, hi)
        const double pl = phil + (double)lo, pr = phir + (double)lo;
        const double cl = rcp_nr1(pl) * taper_upper(hl * pl);                        // c(n) decreases in n: its value at the range's start
        const double cr = rcp_nr1(pr) * taper_upper(hr * pr);
        tail = fma(cl, abs_range_bound(p16, nblk, offset + nr - (hi - 1), offset + nr - lo), tail);
        tail = fma(cr, abs_range_bound(p16, nblk, offset + nl + lo, offset + nl + hi - 1), tail);
    }
    tail *= s0;
    const double ub = head + tail * (1.0 + 1.0e-9) + (1.0e-9 + 1.0e-11 * p16[nblk]);   // rounding of either sum is far below this
    return (ub != ub) ? INF : ub;
}

// The same bound with one candidate's work spread over FOUR consecutive lanes (sub = lane & 3): the head terms and the
// tail ranges are dealt round-robin and summed over the quad.  Frames with few candidates (voiced speech: ~10) would
// otherwise run the whole bound on a fifth of the lanes.  The sums are associated differently from first_eval_bound;
// both are upper bounds with the same margin, and only the ORDER and the amount of refinement work depend on them.
// Must be called from converged code (every lane of the quad, `have` = the quad holds a candidate).
__device__ __forceinline__ double first_eval_bound_quad(const double *ys, const double *p16, int nblk, int nvalid, int ylen,
                                                        int offset, int nx, double nn, int depth, int sub, bool have) {
    const double INF = __builtin_inf();
    bool inf = !have;
    double v0 = 0.;
    if (!inf) {
        if (nn == 0. || nn >= (double)nx || !(nn - 1. < nn + 1.)) inf = true;
        else {
#pragma clang fp contract(off)
            const double golden = 1. - 0.6180339887498948482045868343656381177203091798057628621;
            const double ba = nn - 1., bb = nn + 1.;
            if (!(ba >= (double)(-offset))) inf = true;
            v0 = ba + golden * (bb - ba);
        }
    }
    if (!inf && (nx < 1 || v0 > (double)nx || v0 < 0.0)) inf = true;
    int nl = 0, nr = 1, D = 0;
    double phil = 0.5, phir = 0.5;
    if (!inf) {
        const double fl = floor(v0);
        nl = (int)fl; nr = nl + 1;
        phil = v0 - (double)nl; phir = 1.0 - phil;
        if (fabs(v0 - (double)nl) < 1.0e-10 || fabs(v0 - (double)nr) < 1.0e-10) inf = true;
        D = depth;
        if ((offset + nr) < D) D = ((offset + nr) < 0) ? 0 : (offset + nr);            // :46-52
        if ((offset + nl + D) >= nx) D = nx - offset + nl - 1;                          // :55-57
        if (D < 0 || offset + nr >= ylen || D > offset + nr || offset + nl < 0 || offset + nl + D >= ylen) inf = true;
    }
    double head = 0.0, tail = 0.0;
    if (!inf) {
        const double s0 = sin_poly(M_PI * fmin(phil, phir)) * 0.31830988618379067154;   // |sin(pi ph)| / pi >= 0
        const double hl = M_PI * rcp_nr1(phil + (double)D), hr = M_PI * rcp_nr1(phir + (double)D);
        const int nh = (BOUND_HEAD < D + 1) ? BOUND_HEAD : D + 1;
        for (int m = sub; m < nh; m += 4) {
            const double pl = phil + (double)m, pr = phir + (double)m;
            const double tl = y_at(ys, nvalid, offset + nr - m) * rcp_nr1(pl) * fma(0.5, cos_0_pi(hl * pl), 0.5);
            const double tr = y_at(ys, nvalid, offset + nl + m) * rcp_nr1(pr) * fma(0.5, cos_0_pi(hr * pr), 0.5);
            const double t = tl + tr;
            head += (m & 1) ? -t : t;
        }
        head *= s0;
        int r = 0;
        for (int lo = nh; lo <= D; lo *= 2, r++) {
            if ((r & 3) != sub) continue;
            const int hi = (2 * lo < D + 1) ? 2 * lo : D + 1;                            // terms [lo, hi)
            const double pl = phil + (double)lo, pr = phir + (double)lo;
            const double cl = rcp_nr1(pl) * taper_upper(hl * pl);
            const double cr = rcp_nr1(pr) * taper_upper(hr * pr);
            tail = fma(cl, abs_range_bound(p16, nblk, offset + nr - (hi - 1), offset + nr - lo), tail);
            tail = fma(cr, abs_range_bound(p16, nblk, offset + nl + lo, offset + nl + hi - 1), tail);
        }
        tail *= s0;
    }
    head = group_sum<4>(head);
    tail = group_sum<4>(tail);
    const double ub = head + tail * (1.0 + 1.0e-9) + (1.0e-9 + 1.0e-11 * p16[nblk]);   // rounding of the sums is far below this
    return (inf || ub != ub) ? INF : ub;
}

// index of the largest key >= bar among keys[0, ncand) (lowest index on ties), or -1; the winner is retired
__device__ __forceinline__ int pick_best(float *keys, int ncand, double bar, int lane) {
    double bv = -__builtin_inf(); int bi = 0x7fffffff;
    for (int i = lane; i < ncand; i += 64) { const double v = (double)keys[i]; if (v > bv) { bv = v; bi = i; } }
    const double gm = wave_max(bv);
    int pick = (bv == gm) ? bi : 0x7fffffff;
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(pick, o, 64); pick = (other < pick) ? other : pick; }
    if (pick == 0x7fffffff || !(gm >= bar)) return -1;
    if (lane == 0) keys[pick] = -__builtin_inff();
    wave_sync();
    return pick;
}


// The same eligibility (bound >= bar), another ORDER: among the eligible candidates the one whose refinement will most
// likely end highest.  Brent minimises the mirrored interpolant (Q6 + Q8), which falls towards the integer lag k from both
// sides with limits y[k-1] and y[k+1]: a refinement usually ends at the smaller neighbour sample, well below the bound f(v0)
// it started from.  Taking the candidate with the largest min(y[k-1], y[k+1]) first raises the bar to (nearly) its final
// height at once, and the competitors -- whose bounds may exceed the winner's final strength -- are retired after a few
// evaluations instead of being refined to the end first (replay on the oracle, 172 voiced frames: 28.2 -> 25.3 evaluations
// per frame; refining the best candidate alone takes 24.7).  The order changes nothing that is returned: a candidate that
// is never picked has a bound below the final bar, one that is abandoned has a running minimum below it.
__device__ __forceinline__ int pick_best_pred(float *keys, const cand_t *cand_list, const double *ys, int ncand, double bar, int lane) {
    double bv = -__builtin_inf(); int bi = 0x7fffffff;
    for (int i = lane; i < ncand; i += 64) {
        const float kv = keys[i];                             // -inf: retired (the bar itself is -inf until kmax are kept)
        if (kv > -__builtin_inff() && (double)kv >= bar) {
            const int k = (int)cand_list[i];
            double p = fmin(ys[k - 1], ys[k + 1]);
            if (!(p == p)) p = __builtin_inf();               // NaN data: such a candidate is refined like any other
            if (bi == 0x7fffffff || p > bv) { bv = p; bi = i; }
        }
    }
    const double gm = wave_max(bv);
    int pick = (bi != 0x7fffffff && bv == gm) ? bi : 0x7fffffff;
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(pick, o, 64); pick = (other < pick) ? other : pick; }
    if (pick == 0x7fffffff) return -1;
    if (lane == 0) keys[pick] = -__builtin_inff();
    wave_sync();
    return pick;
}


// LDS of the refinement, in this order from `ys`:  y[n + Y_PAD] | p16[nblk + 1 (+pad)] | keys[n/4 + 8] (float) |
// candidate list (uint16).  pitch_refine_lds_doubles(n) is that footprint in doubles (rounded up).
// nst: lags of the curve that are stored (pitch_curve_entries; 0 = all n)
// cap: entries of keys / candidate list when the list arrives filtered (pitch_refine_store<2>), 0 = n/4 + 8
__host__ __device__ constexpr int pitch_refine_lds_bytes(int n, int nst = 0, int cap = 0) {
    const int m = (nst > 0) ? nst : n;
    return (m + Y_PAD + ((((m + PB - 1) / PB) + 2) & ~1)) * 8 + ((cap > 0) ? cap : n / 4 + 8) * (int)(sizeof(float) + sizeof(cand_t));
}

// How much of the lag curve the refinement can read, when that is less than all of it (0 = all n lags).  The peak scan looks
// at lags [0, n/2] (src/periodic.rs:416-419).  A candidate passes the frequency filter (:439) with a parabolic lag below
// sample_rate / fmin; its Brent bracket reaches one lag further, its sinc sum as deep as the samples on its left (:46-52):
// the right-most lag of any term is < 2 sample_rate / fmin + 4.  (A candidate of :194 reads lag nx - 1 = n for an even n:
// past the data, zero, like everything from the cut on.)  For long frames at speech settings this is half of the curve:
// at 4,096 samples / 48 kHz / fmin 75 Hz lags [0, 2050), and the frame state in LDS shrinks from 46 KB to what the transform's
// exchange buffer needs anyway (35 KB: four frames per CU instead of three); at 2,048 from 22.9 to 17.4 KB (eight per CU, 7).
// the lags a candidate's refinement can read (the first branch of pitch_curve_entries' maximum), 0 = no such bound
// (for an odd n too: its last lag, which pitch_curve_entries keeps the whole curve for, is read by a candidate of :194 only -- one whose
// abscissa lies past lag n/2, i.e. not one the frequency filter lets through from a peak below 2 sr / fmin; the split form sends the others
// to its whole-curve kernel)
inline int pitch_curve_reach(int n, double sample_rate, double fmin) {
    if (!(fmin > 0.0) || !(sample_rate > 0.0)) return 0;
    const double reach = 2.0 * ceil(sample_rate / fmin) + 16.0;
    if (!(reach < (double)n)) return 0;
    return ((int)reach + 1) & ~1;
}
inline int pitch_curve_entries(int n, double sample_rate, double fmin) {
    if ((n & 1) || !(fmin > 0.0) || !(sample_rate > 0.0)) return 0;          // an odd n reads its last lag (:194); NaN: all
    const double reach = 2.0 * ceil(sample_rate / fmin) + 16.0;
    if (!(reach < (double)n)) return 0;
    int m = (int)reach;
    if (m < n / 2 + 2) m = n / 2 + 2;
    m = (m + 1) & ~1;
    return (m < n) ? m : 0;
}

struct pitch_params_t { double sample_rate, threshold, fmin, fmax; int kmax; int full_off; int f32 = 0; int ncurve = 0; };   // full_off: byte offset of
                                                                                                // the full-list region in the
                                                                                                // dynamic LDS, 0 = none, < 0 =
                                                                                                // the frame's output row
// f32 != 0: the T = f32 instantiation of Pitched::pitch (k_f32.hip): every value of type T -- the parabolic lag's frequency,
// the abscissa handed to the refinement, the candidate's frequency and strength -- is rounded to f32 where the generic code
// holds it in T (src/periodic.rs:423-448 with T = f32); nothing is pruned (the exact top-k bounds are stated for f64).
// entries of the full-list region: at most n/4 strict local maxima in [0, n/2) plus the unvoiced candidate
__host__ __device__ constexpr int pitch_full_list_entries(int n) { return n / 4 + 2; }

// Phases a-c of Pitched::pitch on the lag curve ys[0..n) (zero padded to n + Y_PAD; the curve is
// (r / max|r|) / w_lag, src/periodic.rs:404-408), one wavefront per frame; writes the frame's outputs.
// unc_tol > 0: the curve carries an absolute error of up to unc_tol per entry (FFT-based autocorrelation).  Two places
// turn y into a DISCRETE decision from the curve alone: the strict 3-point peak test (:370-374) and the frequency filter
// on the parabolic lag (:439).  If any lag's peak test could come out differently within that error (a peak or a non-peak
// by less than unc_tol, e.g. a curve that is exactly zero over a stretch), or a peak's frequency lies within the error's
// reach of fmin / fmax, nothing is written and the function returns false: the caller hands the frame to the kernel that
// computes the lag sums directly.  (The third discrete decision, the ORDER of the refined strengths, is not decidable
// this way: the Brent iteration is chaotic below its stopping width, so two strengths closer than ~1e-6 can come out in
// either order from ANY summation of the lag sums, the direct one included -- the tests count those as tie swaps.)
// Returns true when the frame's outputs were written.
// full != nullptr (LDS, pitch_full_list_entries(n) entries): the WHOLE Vec of :452-454 is wanted (kmax > 64, more than
// the lane-resident list holds).  Nothing is pruned; every refined candidate is parked at its candidate index, and the
// frame ends with a rank sort by (strength desc, candidate index asc) == the reference's stable sort, whose first kmax
// entries are written.
// STAGE (round 5, the split form of the 4096-point plan): 0 = everything; 1 = the peak scan and the frequency filter only (returns
// after them: *io_ncand candidates, in index order, at the head of the candidate list in LDS -- or at list_at, and then ys may be a
// curve in MEMORY: only the list is written); 2 = everything AFTER them, from a list
// a STAGE-1 call left (*io_ncand entries; keys / list regions of cand_cap entries; ys then only needs the lags a candidate's
// refinement reads, pp.ncurve = pitch_curve_reach).  The same statements in the same order: 1 then 2 is 0, bit for bit.
template <int STAGE = 0>
__device__ __forceinline__ bool pitch_refine_store(double *ys, int n, const pitch_params_t &pp, long f,
                                                   double *__restrict__ out_cand, long cand_ld,
                                                   int32_t *__restrict__ out_count, int32_t *__restrict__ status,
                                                   unsigned long long *__restrict__ work, double unc_tol = 0.0,
                                                   double2 *full = nullptr, int *io_ncand = nullptr, int cand_cap = 0,
                                                   uint16_t *list_at = nullptr) {
    const int lane = lane_id();
    const double sample_rate = pp.sample_rate, threshold = pp.threshold, fmin = pp.fmin, fmax = pp.fmax;
    const int kmax = pp.kmax;
    const bool f32 = pp.f32 != 0;
    const int nst = (pp.ncurve > 0) ? pp.ncurve : n;   // lags stored (pitch_curve_entries): zeros from there on, as past n
    const int nblk = (nst + PB - 1) / PB;            // blocks of PB lags for the |y| prefix sums
    double *p16 = ys + nst + Y_PAD;
    float *keys = reinterpret_cast<float *>(p16 + ((nblk + 2) & ~1));
    // (STAGE 1 reads ys and writes nothing but the list: with list_at the curve may stay where it is, in memory)
    cand_t *cand_list = (STAGE == 1 && list_at != nullptr) ? reinterpret_cast<cand_t *>(list_at)
                                                           : reinterpret_cast<cand_t *>(keys + ((STAGE == 2 && cand_cap > 0) ? cand_cap : n / 4 + 8));
    const int b = (int)floor(0.5 * (double)n);      // brent_ixmax, :414
    const int offset = -b - 1;                      // :429
    const int nx = b - offset;                      // :430
    const int ylen = 2 * n;                         // :411
    const int nvalid = nst + Y_PAD;

    VBX_PHASE_INIT();
    int ncand = 0;
    if constexpr (STAGE == 2) ncand = *io_ncand;
    else {
    // a) peaks -> filtered candidate list.  Two passes so that the two divisions of the frequency filter run once per
    // 64 PEAKS, not once per 64 lags: first every strict local maximum is compacted (index order), then the filter.
    int npeak = 0;
    bool unsure = false;
    for (int base = 0; base < b; base += 64) {
        const int k = base + lane;
        bool peak = false;
        if (k >= 1 && k + 1 < b) {                  // windows(3) over self_lag[0..b] (Q4)
            const double c = ys[k];
            const double d1 = c - ys[k - 1], d2 = c - ys[k + 1];
            peak = (ys[k - 1] < c) && (ys[k + 1] < c);
            unsure = unsure || (fabs((d1 < d2) ? d1 : d2) <= unc_tol);    // NaN and unc_tol = 0 with a clear margin: false
        }
        const unsigned long long mask = __ballot(peak);
        if (peak) cand_list[npeak + __popcll(mask & ((1ull << lane) - 1ull))] = (cand_t)k;
        npeak += __popcll(mask);
    }
    wave_sync();
    if (unc_tol > 0.0 && __any(unsure)) return false;
    VBX_PHASE(work, f, 6);
    unsure = false;
    for (int base = 0; base < npeak; base += 64) {  // in place: the write position never passes the read position
        const int i = base + lane;
        bool pass = false;
        int k = 0;
        if (i < npeak) {
            k = cand_list[i];
            double freq, nn;
            cand_from_peak(ys, k, sample_rate, offset, freq, nn, f32);
            pass = (freq == 0.0) || (freq > fmin && freq < fmax);             // :439
            if (unc_tol > 0.0) {
                // the filter is a discrete decision too: the parabolic lag k + dr / d2r inherits the curve's error (each of
                // the three samples within unc_tol); a frequency that close to fmin or fmax could fall on the other side
                const double c = ys[k], d2r = 2. * c - (ys[k - 1] - ys[k + 1]), dr = 0.5 * (ys[k + 1] - ys[k - 1]);
                const double dlag = unc_tol * (1.0 + 4.0 * fabs(dr / d2r)) / fabs(d2r);            // bound of the lag's error
                const double tol_f = 2.0 * freq * dlag / fabs((double)k + dr / d2r);
                if (!(fabs(freq - fmin) > tol_f && fabs(freq - fmax) > tol_f)) unsure = true;      // NaN: unsure
            }
        }
        wave_sync();                                // all reads of this pass before its writes
        const unsigned long long mask = __ballot(pass);
        if (pass) cand_list[ncand + __popcll(mask & ((1ull << lane) - 1ull))] = (cand_t)k;
        ncand += __popcll(mask);
        wave_sync();
    }

    if (unc_tol > 0.0 && __any(unsure)) return false;         // a frequency within the curve's error of fmin / fmax
    VBX_PHASE(work, f, 7);
    }
    if constexpr (STAGE == 1) { *io_ncand = ncand; return true; }

    // a') first-evaluation bounds.  p16: prefix sums of |y| over blocks of PB; keys[c]: upper bound of candidate c's
    // strength, stored as a float rounded UP (still an upper bound; the whole frame then fits 12 wavefronts per CU)
    // (not when nothing can be pruned -- the whole Vec, a list that never fills, T = f32: the candidates are then taken in
    // index order, below)
    const bool in_order = full != nullptr || kmax > ncand || f32;
    if (!in_order) {
        // lane l owns the consecutive blocks [l*per, (l+1)*per): local sums, one scan over the lanes, prefix written back
        const int per = (nblk + 63) >> 6;
        double tot = 0.0;
        for (int q = 0; q < per; q++) {
            const int j = lane * per + q;
            if (j < nblk) tot += block_abs_sum(ys + PB * j);   // entries past n are zero
        }
        const double incl = wave_inclusive_scan(tot);        // inclusive scan over the lanes (DPP: no LDS round trips)
        double run = incl - tot;
        if (lane == 0) p16[0] = 0.0;
        for (int q = 0; q < per; q++) {
            const int j = lane * per + q;
            if (j < nblk) {
                run += block_abs_sum(ys + PB * j);
                p16[j + 1] = run;
            }
        }
        wave_sync();
        // one lane per candidate, or -- when that would leave most lanes idle -- four lanes per candidate
        // (cost model: 16 candidates per pass at about 7/25 of the cost of a 64-candidate pass)
        const bool quad = ((ncand + 15) / 16) * 7 < ((ncand + 63) / 64) * 25;
        if (quad) {
            for (int base = 0; base < ncand; base += 16) {
                const int c = base + (lane >> 2);
                const bool have = c < ncand;
                double freq = 0., nn = 0.;
                if (have) cand_from_peak(ys, cand_list[c], sample_rate, offset, freq, nn, f32);
                const double ub = first_eval_bound_quad(ys, p16, nblk, nvalid, ylen, offset, nx, nn, 1200, lane & 3, have);
                const double kb = (ub <= 1.) ? ub : ((ub != ub) ? __builtin_inf() : ((ub == __builtin_inf()) ? ub : 1.));
                if (have && (lane & 3) == 0) keys[c] = __double2float_ru(kb);
            }
        } else {
            for (int c = lane; c < ncand; c += 64) {
                double freq, nn;
                cand_from_peak(ys, cand_list[c], sample_rate, offset, freq, nn, f32);
                const double ub = first_eval_bound(ys, p16, nblk, nvalid, ylen, offset, nx, nn, 1200);
                const double kb = (ub <= 1.) ? ub : ((ub != ub) ? __builtin_inf() : ((ub == __builtin_inf()) ? ub : 1.));
                keys[c] = __double2float_ru(kb);
            }
        }
        wave_sync();
    }

    VBX_PHASE(work, f, 8);
    int st = 0;
    int kept = 0;
    double lf = 0.0, ls = 0.0;                      // lane j holds sorted candidate j
    bool any_nan = false;
    int li = 0;                                     // candidate index of the list entry held by this lane
    // maxima.push(Pitch::new(0, threshold)) (:452) carries the largest index; it enters the list first so
    // that the pruning bar below is armed from the start
    { lf = 0.0; ls = threshold; li = ncand; kept = 1; }
    const bool fullm = full != nullptr;
    if (fullm && lane == 0) full[ncand] = double2{0.0, threshold};

    // Exact top-k pruning.  The caller asked for the first kmax entries of the sorted list.  brent_maximize only
    // ever replaces fx by a value <= fx (:162), so a candidate's final strength is <= min(f(v0), 1): v0 is its
    // first abscissa and the "> 1 -> 1/s" reflection of :446 keeps strengths <= 1.  A candidate whose bound
    // (keys[], or the exact f(v0) once evaluated) is strictly below the kmax-th best strength already in the
    // list cannot be among the entries returned, and is skipped.  Candidates are taken best-bound-first, so the
    // bar rises as early as possible; when the best remaining bound is below the bar, all the rest is too.
    // A list that never fills (kmax >= count) keeps bar = -inf: everything is refined.
#define VBX_BAR() ((!fullm && !f32 && kept == kmax) ? readlane_f64(ls, kmax - 1) : -__builtin_inf())
    auto insert = [&](double f_g, double s_g, int c_g) {
        if (s_g != s_g) any_nan = true;
        if (fullm) { if (lane == 0) full[c_g] = double2{f_g, s_g}; return; }
        const int pos = __popcll(__ballot(lane < kept && (ls > s_g || (ls == s_g && li < c_g))));
        const double pf = from_prev_lane(lf), ps = from_prev_lane(ls);
        const int pi = __builtin_amdgcn_update_dpp(0, li, DPP_WAVE_SHR1, 0xf, 0xf, true);
        if (lane > pos) { lf = pf; ls = ps; li = pi; }
        if (lane == pos) { lf = f_g; ls = s_g; li = c_g; }
        kept = (kept + 1 < kmax) ? kept + 1 : kmax;
    };
    unsigned cterms = 0, cevals = 0;                // sinc terms / evaluations executed (uniform)

    // b, c) refinement, most promising candidate first (pick_best_pred), one at a time with all 64 lanes on its sinc sums
    // (improve_extremum, :192-229): every lane runs the reference's Brent iteration on identical values (the wave
    // sums are bit-identical in all lanes).  In a voiced frame the first strength becomes the bar that retires every
    // other candidate without an evaluation.  Finished candidates enter the lane-resident list ordered by
    // (strength desc, candidate index asc) == the reference's stable sort (:453).
    // Frames with many candidates (noise-like frames: short lags, little work per evaluation) use four groups of 16
    // lanes instead, each refining one candidate and taking the next-best as soon as its own has converged.  So do frames
    // with at least four candidates when the caller keeps four or more (kmax >= 4): several candidates have to be refined
    // to the end whatever the bar does, and four at a time share the per-evaluation overhead (+9 % at kmax = 8, +8 % at 64).
    // A candidate's result does not depend on which others are refined, but its last bits depend on which path summed its
    // sinc terms (64 or 16 lanes): lists returned for kmax in {1, 2, 3} are bit for bit the head of one another, and so are
    // the lists for every kmax >= 4; between the two classes a candidate agrees within the Brent iteration's own scatter
    // (~1e-7 relative in Hz), counts and statuses exactly (tests/test_gpu_parity.py::test_pitch_topk_is_the_prefix...).
    // Nothing can be pruned (in_order: the whole Vec, or fewer candidates than the caller keeps): every candidate is refined
    // to the end whatever the others do, so the order is free, and it is the order of DESCENDING lag: a candidate at lag k sums 2 (k + 2) sinc terms per evaluation (:46-52 clamp the depth to
    // the samples on its left), 160 .. 1,200 over the searched lags.  The four groups of a wavefront step through their
    // evaluations together, so each round lasts as long as its longest sum: with neighbours in lag side by side the four
    // sums are within a few per cent of one another (taken by bound, i.e. in no particular order of lag, the longest of four
    // is ~1.45x the mean), and with the longest first the frame's tail -- groups idle because the list has run out -- is
    // made of the shortest.
    int next_full = ncand - 1;
    unsigned nterms = 0, nevals = 0;                // group path: work executed (group leaders' counts are summed)
    int group_lanes = RG;
    if (ncand <= GROUP_PATH_MIN_CAND && !(kmax >= VBX_EXP_GROUP_KMAX && ncand >= 4)) {
        for (;;) {
            const double bar = VBX_BAR();
            const int c = in_order ? next_full-- : pick_best_pred(keys, cand_list, ys, ncand, bar, lane);
            if (c < 0) break;
            double freq, nn, xmid, ymid;
            cand_from_peak(ys, cand_list[c], sample_rate, offset, freq, nn, f32);
            bool dropped = false;
#ifndef VBX_EXP_NO_WAVE_BRENT
            if (!improve_extremum_sinc_wave(ys, nvalid, ylen, offset, nx, nn, 1200, xmid, ymid, cterms, cevals, bar, dropped))
#endif
                improve_extremum_sinc<64>(ys, nvalid, ylen, offset, nx, nn, 1200, true, xmid, ymid, st, &cterms, &cevals, bar, &dropped);
            if (dropped) continue;
            double xm, ym;
            {
#pragma clang fp contract(off)
                xm = xmid + (double)offset;                                   // :445
                ym = ymid;
                if (ym > 1.) ym = 1. / ym;                                    // :446
                xm = sample_rate / (f32 ? (double)(float)xm : xm);            // :447
                if (f32) { xm = (double)(float)xm; ym = (double)(float)ym; }  // Pitch<f32>
            }
            insert(xm, ym, c);
        }
    } else {
      // PG lanes per candidate; DUAL: each of them stands in for two lanes of a group of 2 PG (sinc_terms_fast_dual)
      auto run_groups = [&](auto pg_tag, auto dual_tag) {
        constexpr int PG = decltype(pg_tag)::value;
        constexpr bool DUAL = decltype(dual_tag)::value;
        bool exhausted = false;
        const int gid = lane / PG;
        int ci = -1, it = 0;
        bool special = false, safe = false, trusted = false;
        double ba = 0., bb = 0., v = 0., w = 0., x = 0., fv = 0., fw = 0., fx = 0., xmid = 0., ymid = 0.;
        constexpr unsigned long long LEADERS = (PG == 16) ? 0x0001000100010001ull : (PG == 8) ? 0x0101010101010101ull
                                             : (PG == 32) ? 0x0000000100000001ull : (PG == 4) ? 0x1111111111111111ull : 1ull;
        const double golden = 1. - 0.6180339887498948482045868343656381177203091798057628621;
        const double sqrt_epsilon = 1.4901161193847656e-08, eps = 2.220446049250313e-16, tol = 1e-10;
        for (;;) {
#pragma clang fp contract(off)   // the scalar Brent arithmetic stays bit-identical to the unfused CPU arithmetic
            // 1. which groups' candidates have converged (:131-137)?  They enter the list and their groups take new candidates
            // in the SAME round: a group never sits through a round of the others' sinc sums without one of its own.
            bool finished = false;
            double range = 0., middle_range = 0., tol_act = 0.;
            if (ci >= 0) {
                if (special) finished = true;
                else if (it > 0) {
                    range = bb - ba;
                    middle_range = (ba + bb) * 0.5;
                    tol_act = sqrt_epsilon * fabs(x) + tol / 3.;
                    if (it > 60 || fabs(x - middle_range) + range * 0.5 <= 2. * tol_act) { finished = true; xmid = x; ymid = fx; }
                }
            }
            if (__any(finished)) {                               // finished candidates -> sorted list
                unsigned long long fm = __ballot(finished) & LEADERS;
                double xm = xmid + (double)offset;                                // :445
                double ym = ymid;
                if (ym > 1.) ym = 1. / ym;                                        // :446
                double cf = sample_rate / (f32 ? (double)(float)xm : xm), cs = ym;   // :447-448
                if (f32) { cf = (double)(float)cf; cs = (double)(float)cs; }      // Pitch<f32>
                while (fm) {
                    const int ld = __builtin_ctzll(fm);
                    fm &= fm - 1;
                    insert(readlane_f64(cf, ld), readlane_f64(cs, ld), __builtin_amdgcn_readlane(ci, ld));
                }
                if (finished) ci = -1;
            }
            {   // 2. hand the next candidates to the idle groups, in group order
                unsigned long long im = __ballot(ci < 0) & LEADERS;
                while (im != 0ull && !exhausted) {
                    const int c = in_order ? next_full-- : pick_best_pred(keys, cand_list, ys, ncand, VBX_BAR(), lane);
                    if (c < 0) { exhausted = true; break; }
                    const int g = __builtin_ctzll(im) / PG;
                    im &= im - 1ull;
                    if (gid == g) {
                        ci = c;
                        double freq, nn;
                        cand_from_peak(ys, cand_list[ci], sample_rate, offset, freq, nn, f32);
                        it = 0; special = false; xmid = 0.; ymid = 0.;
                        if (nn == 0.) { special = true; xmid = 0.; ymid = ys[0]; }                              // :193
                        else if (nn >= (double)nx) { special = true; xmid = (double)nx; ymid = y_at(ys, nvalid, nx - 1); }   // :194
                        else if (!(nn - 1. < nn + 1.)) { special = true; st |= 4; }                             // assert!(a < b), :113
                        ba = nn - 1.; bb = nn + 1.;
                        safe = ba >= (double)(-offset);      // no abscissa of the bracket can index out of bounds
                        trusted = !special && sinc_bracket_trusted(ba, bb, nvalid, ylen, offset, nx, 1200);
                    }
                }
            }
            if (!__any(ci >= 0)) break;
            const double bar = VBX_BAR();

            // 3. every group's next abscissa (:138-155; a candidate that came in above starts at the golden section, :114), the
            // sinc sums of all groups together, and the bookkeeping of :157-186.  (A candidate of :193-194 takes no
            // evaluation; it is entered in the next round's step 1.)
            bool need = false;
            double t = 0.;
            if (ci >= 0 && !special) {
                if (it == 0) { v = ba + golden * (bb - ba); t = v; }
                else {
                    double new_step = (x < middle_range) ? golden * (bb - x) : golden * (ba - x);
                    if (fabs(x - w) >= tol_act) {
                        const double tt = (x - w) * (fx - fv);
                        double q = (x - v) * (fx - fw);
                        double pp = (x - v) * q - (x - w) * tt;
                        q = 2. * q - tt;
                        if (q > 0.) pp = -pp; else q = -q;
                        if (fabs(pp) < fabs(new_step * q) && pp > q * (ba - x + 2. * tol_act) && pp < q * (bb - x - 2. * tol_act))
                            new_step = pp / q;
                    }
                    if (fabs(new_step) < tol_act) new_step = (new_step > 0.) ? tol_act : -tol_act;
                    t = x + new_step;
                }
                need = true;
            }
            const double ft = sinc_interp<PG, DUAL>(ys, nvalid, ylen, offset, nx, t, 1200, need, st, &nterms, trusted);
            nevals += need ? 1u : 0u;
            if (need) {
                if (it == 0) {
                    x = v; w = v; fv = ft; fx = ft; fw = ft; it = 1;
                    const double ub = (ft <= 1.) ? ft : 1.;          // NaN -> 1: never pruned
                    if (ub < bar && safe) ci = -1;                   // pruned: it cannot be among the entries returned
                } else {
                    if (ft <= fx) {
                        if (t < x) bb = x; else ba = x;
                        v = w; w = x; x = t;
                        fv = fw; fw = fx; fx = ft;
                    } else {
                        if (t < x) ba = t; else bb = t;
                        if (ft <= fw || fabs(w - x) < eps) {
                            v = w; w = t;
                            fv = fw; fw = ft;
                        } else if (ft <= fv || fabs(v - x) < eps || fabs(v - w) < eps) {
                            v = t;
                            fv = ft;
                        }
                    }
                    it++;
                    // the final strength is <= the current fx (:162): strictly below the bar it cannot be returned
                    if (safe && fx < bar) ci = -1;
                }
            }
        }
      };
      // Nothing can be pruned and there are candidates for more than four groups: eight groups of RG / 2 lanes, each lane
      // standing in for two -- bit for bit the sums of the RG-lane groups (so the lists of every kmax >= 4 stay the head of
      // one another), with the bookkeeping of a round shared by eight candidates instead of four.
      // (also where the bar can rise, from kmax = 8 and twelve candidates on: a few more evaluations are started on candidates
      // the bar then retires, 248 -> 259 per frame at kmax = 8, and the rounds are still cheaper: 6.41 -> 6.50, kmax 64: 3.15 -> 3.28 M)
      const bool dual_ok = in_order || (kmax >= 8 && ncand >= 12);
      if (RG == 16 && dual_ok && ncand >= DUAL_MIN_CAND) { run_groups(std::integral_constant<int, RG / 2>{}, std::true_type{}); group_lanes = RG / 2; }
      else run_groups(std::integral_constant<int, RG>{}, std::false_type{});
    }
#undef VBX_BAR
    VBX_PHASE(work, f, 9);
    const int total_cand = ncand + 1;
    st = __any(st & 4) ? 4 : 0;
    if (total_cand > 1 && (any_nan || threshold != threshold)) st |= 8;   // partial_cmp().unwrap() panics (Q10)
    int code = 0;
    if (st & 4) code = 4; else if (st & 8) code = 3;
    if (fullm) {
        wave_sync();
        if (pp.full_off < 0) {
            // the candidates were parked in the frame's own output row (global memory; the launch guarantees that the
            // row holds the whole Vec): bring them into LDS, over the lag curve that nobody reads any more, and sort there
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            double2 *lds_list = reinterpret_cast<double2 *>(ys);
            for (int i = lane; i < total_cand; i += 64) lds_list[i] = full[i];
            full = lds_list;
            wave_sync();
        }
        double *row = out_cand + f * cand_ld;
        const int total = (code == 0) ? total_cand : 0;
        for (int i0 = 0; i0 < total; i0 += 64) {
            const int i = i0 + lane;
            const double2 me = (i < total) ? full[i] : double2{0.0, 0.0};
            int rank = 0;
            for (int j = 0; j < total; j++) {
                const double sj = full[j].y;                 // same address in every lane: one broadcast read
                rank += (sj > me.y || (sj == me.y && j < i)) ? 1 : 0;
            }
            if (i < total && rank < kmax) *reinterpret_cast<double2 *>(row + 2 * rank) = me;
        }
        for (int i = total + lane; i < kmax; i += 64) *reinterpret_cast<double2 *>(row + 2 * i) = double2{0.0, 0.0};
    } else if (lane < kmax) {
        const bool valid = (code == 0) && lane < kept;
        double2 o;
        o.x = valid ? lf : 0.0;                     // Pitch { frequency, strength }
        o.y = valid ? ls : 0.0;
        *reinterpret_cast<double2 *>(out_cand + f * cand_ld + 2 * lane) = o;   // row f of a [F, cand_ld] array of doubles
    }
    if (lane == 0) {
        if (out_count != nullptr) out_count[f] = (code == 0) ? total_cand : 0;
        if (status != nullptr) status[f] = code;
    }
    if (work != nullptr) {                          // profiling only: frames, candidates, sinc evaluations, sinc terms
        const bool leader = (lane & (group_lanes - 1)) == 0;
        unsigned long long te = leader ? nterms : 0u, ev = leader ? nevals : 0u;
        for (int o = 32; o > 0; o >>= 1) { te += __shfl_xor(te, o, 64); ev += __shfl_xor(ev, o, 64); }
        if (lane == 0) {
            unsigned long long *w = work + 4 * (f & (PITCH_WORK_SLOTS - 1));
            atomicAdd(w + 0, 1ull); atomicAdd(w + 1, (unsigned long long)ncand);
            atomicAdd(w + 2, ev + cevals); atomicAdd(w + 3, te + cterms);
        }
    }
    return true;
}

}  // namespace vbx
