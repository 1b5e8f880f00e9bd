// vbx_device.hpp -- wave64 device helpers shared by the gfx950 kernels.
// CDNA4 only: 64-lane wavefronts, DPP row rotations, v_readlane broadcasts.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define VBX_WAVE 64

namespace vbx {

// ---- lane broadcast / shuffles on f64 (two 32-bit halves) ---------------------------

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readfirstlane_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}

// DPP move of one f64 (both halves).  CTRL is a gfx9 dpp_ctrl immediate; lanes whose
// source is out of range receive 0 (bound_ctrl).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;   // lane 15 of each row -> the next row; lane 31 -> rows 2, 3

constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;
constexpr int DPP_WAVE_SHL1 = 0x130;  // lane i <- lane i+1 (lane 63 <- 0)
constexpr int DPP_WAVE_SHR1 = 0x138;  // lane i <- lane i-1 (lane 0 <- 0)

// value of lane+1 (lane 63 receives 0)
__device__ __forceinline__ double from_next_lane(double v) { return dpp_f64<DPP_WAVE_SHL1>(v); }
// value of lane-1 (lane 0 receives 0)
__device__ __forceinline__ double from_prev_lane(double v) { return dpp_f64<DPP_WAVE_SHR1>(v); }

// A value the compiler must treat as computed: stops FP contraction from fusing the multiplication that produced it into
// a following addition.  A product that feeds a cross-lane sum must be ROUNDED before the exchange -- fused as
// fma(a_i, k_i, dpp(round(a_j k_j))) in lane i and fma(a_j, k_j, dpp(round(a_i k_i))) in lane j, the two partners of a
// butterfly step end up an ulp apart, and the "bit-identical in every lane of the group" property of group_sum is gone
// (whether the compiler fuses depends on the inlining context: the same source gave different bits in two kernels).
__device__ __forceinline__ double rounded(double v) { asm volatile("" : "+v"(v)); return v; }

// Sum over the 64 lanes; the result is bit-identical in every lane (it is broadcast from
// scalar registers), which keeps data-dependent control flow wave-uniform.
__device__ __forceinline__ double wave_sum(double v) {
    v = rounded(v);
    // all-reduce inside each 16-lane row by rotations (full-rate DPP moves)
    v += dpp_f64<DPP_ROW_ROR8>(v);
    v += dpp_f64<DPP_ROW_ROR4>(v);
    v += dpp_f64<DPP_ROW_ROR2>(v);
    v += dpp_f64<DPP_ROW_ROR1>(v);
    // one lane per row -> scalar registers -> uniform sum
    double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
    return (a + b) + (c + d);
}

// Inclusive prefix sum over the 64 lanes, all on the vector ALU (no LDS crossbar: six ds_bpermute round trips otherwise): a
// row-level scan by row_shr:1, 2, 4, 8 (a lane without a source adds 0), then the totals of the rows before -- lane 15 of a
// row into the next row (rows 1 and 3), lane 31 into rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64_or_zero(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);   // rows outside ROW_MASK, lanes without a source: 0
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_inclusive_scan(double v) {
    v += dpp_f64<0x111>(v);                      // row_shr:1
    v += dpp_f64<0x112>(v);                      // row_shr:2
    v += dpp_f64<0x114>(v);                      // row_shr:4
    v += dpp_f64<0x118>(v);                      // row_shr:8
    v += dpp_f64_or_zero<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_f64_or_zero<DPP_ROW_BCAST31, 0xC>(v);
    return v;
}

__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_f64<DPP_ROW_ROR8>(v));
    v = fmax(v, dpp_f64<DPP_ROW_ROR4>(v));
    v = fmax(v, dpp_f64<DPP_ROW_ROR2>(v));
    v = fmax(v, dpp_f64<DPP_ROW_ROR1>(v));
    double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
    return fmax(fmax(a, b), fmax(c, d));
}

// reference-model variants through ds_bpermute (used by the self-test to validate the DPP forms)
__device__ __forceinline__ double wave_sum_shfl(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return readfirstlane_f64(v);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// Workgroup -> item mapping for one-wavefront workgroups over a hop-strided frame view.  Workgroups are dealt round-robin
// to the 8 XCDs (workgroup b runs on XCD b % 8), each with its own L2; neighbouring frames share frame_len - hop of their
// samples (60 % at 25 ms / 10 ms), so with the identity mapping every XCD fetches those samples from HBM again.  Inside a
// tile of 8 * XCD_RUN items, XCD x takes the XCD_RUN CONSECUTIVE items [x * XCD_RUN, (x + 1) * XCD_RUN) in dispatch order:
// the overlap is then served by that XCD's L2.  The incomplete last tile keeps the identity mapping.
constexpr long XCD_RUN = 1024;
__device__ __forceinline__ long xcd_item(long b, long n) {
    constexpr long T = 8 * XCD_RUN;
    const long tile = b / T;
    if ((tile + 1) * T > n) return b;
    const long j = b - tile * T;
    return tile * T + (j & 7) * XCD_RUN + (j >> 3);
}

constexpr int DPP_QUAD_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_REV = 0x1B;          // quad_perm [3,2,1,0]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // i <-> 7-i inside each 8 lanes
constexpr int DPP_ROW_MIRROR = 0x140;       // i <-> 15-i inside each 16 lanes

// Sum over each group of G lanes.  Every step pairs lanes by an involution, and fp addition is
// commutative, so all lanes of a group end with bit-identical results: data-dependent control flow
// stays uniform inside a group.  Must be called from converged code.
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    static_assert(G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group size");
    v = rounded(v);                              // a product that arrives here is rounded before the first exchange
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_REV>(v);
    if (G >= 8) v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    if (G >= 16) v += dpp_f64<DPP_ROW_MIRROR>(v);
    if (G == 64) {
        // every lane of row r holds that row's sum (a, b, c, d for rows 0..3); wanted: (a + b) + (c + d) in every lane.
        // Two wave-level DPP broadcasts -- lane 15 of a row into the next row (rows 1 and 3 take it: b + a, d + c), lane 31
        // into rows 2 and 3 (row 3: (d + c) + (b + a)) -- and ONE readlane pair: the same three additions (fp addition is
        // commutative: the same bits) in 10 instructions instead of eight readlanes, four moves and three additions.
        // (every row is written -- rows without a source lane receive 0, row 2 receives sums nobody reads -- so the moves
        // need no copy of the old value; only lane 63 is read)
        v += dpp_f64<DPP_ROW_BCAST15>(v);
        v += dpp_f64<DPP_ROW_BCAST31>(v);
        return readlane_f64(v, 63);
    }
    if (G == 32) {
        const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
        return (lane_id() < 32) ? (a + b) : (c + d);
    }
    return v;
}

// Orders LDS traffic between the lanes of ONE wavefront (LDS executes a wave's instructions in
// order; this only stops the compiler from moving accesses across the point).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// reciprocal of a normal, well-scaled double: v_rcp_f64 (4.6e-8) + Newton steps
__device__ __forceinline__ double rcp_nr1(double a) {       // ~2e-15 relative
    double r = __builtin_amdgcn_rcp(a);
    return fma(fma(-a, r, 1.0), r, r);
}
__device__ __forceinline__ double rcp_nr2(double a) {       // correctly rounded in practice
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    return fma(fma(-a, r, 1.0), r, r);
}

// x / denom for |x| <= 32768: reciprocal multiply + one Markstein correction step.  Correctly rounded
// (== true division, which is what the reference computes) -- checked exhaustively over all 65536
// int16 values in tests/test_gpu_frontend.py.
__device__ __forceinline__ double div_exact_small(double x, double denom, double r) {
    const double q = x * r;
    const double rem = fma(-q, denom, x);
    return fma(rem, r, q);
}
// 16-bit PCM sample -> f64 as the reference's callers ingest WAV data (hound: i16 as f64 / 32767, tests/lib.rs:17-19):
// what vbx_pcm16_to_f64 writes, computed in registers by the kernels that take PCM frames directly
__device__ __forceinline__ double pcm16_value(int s) { return div_exact_small((double)s, 32767.0, 1.0 / 32767.0); }

// ---- complex arithmetic (num-complex 0.2 formulas; FMA contraction allowed) -------
// Generic over the scalar: Complex<f64> on the hot path, Complex<f32> for the f32 instantiation of Polynomial
// (src/polynomial.rs:336-386).

template <typename T> struct cx { T re, im; };
using c64 = cx<double>;
using c32 = cx<float>;

template <typename T> __device__ __forceinline__ cx<T> cmk(T re, T im) { cx<T> z; z.re = re; z.im = im; return z; }
template <typename T> __device__ __forceinline__ cx<T> cadd(cx<T> a, cx<T> b) { return cmk<T>(a.re + b.re, a.im + b.im); }
template <typename T> __device__ __forceinline__ cx<T> csub(cx<T> a, cx<T> b) { return cmk<T>(a.re - b.re, a.im - b.im); }
template <typename T> __device__ __forceinline__ cx<T> cneg(cx<T> a) { return cmk<T>(-a.re, -a.im); }
template <typename T> __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
    return cmk<T>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
// a*b + c
template <typename T> __device__ __forceinline__ cx<T> cmad(cx<T> a, cx<T> b, cx<T> c) {
    return cmk<T>(fma(a.re, b.re, fma(-a.im, b.im, c.re)), fma(a.re, b.im, fma(a.im, b.re, c.im)));
}
// 1 / |b|^2 of cdiv.  f64: v_rcp_f64 + two Newton steps (correctly rounded in practice) instead of the IEEE division
// sequence -- 7 instructions fewer, three times per Laguerre iteration of a vector-issue-bound kernel.  The two differ only
// where |b|^2 is 0, inf or denormal (NaN instead of inf / a huge number): there the reference's own iterate is already
// non-finite or frozen (a converged lane ignores the step), and no resonance comes out of such a root either way.
__device__ __forceinline__ double cdiv_recip(double ns) {
#ifdef VBX_IEEE_CDIV
    return 1.0 / ns;
#else
    return rcp_nr2(ns);
#endif
}
__device__ __forceinline__ float cdiv_recip(float ns) { return 1.0f / ns; }
template <typename T> __device__ __forceinline__ cx<T> cdiv(cx<T> a, cx<T> b) {
    T ns = b.re * b.re + b.im * b.im;
    T inv = cdiv_recip(ns);
    T re = a.re * b.re + a.im * b.im;
    T im = a.im * b.re - a.re * b.im;
    return cmk<T>(re * inv, im * inv);
}
template <typename T> __device__ __forceinline__ T cnorm(cx<T> a) { return hypot(a.re, a.im); }
template <typename T> __device__ __forceinline__ bool ciszero(cx<T> a) { return a.re == T(0) && a.im == T(0); }
// principal square root, algebraic form (equals the polar form of num-complex up to rounding)
template <typename T> __device__ __forceinline__ cx<T> csqrt(cx<T> z) {
    T r = hypot(z.re, z.im);
    if (r == T(0)) return cmk<T>(T(0), z.im);
    if (z.re >= T(0)) {
        T t = sqrt(T(0.5) * (r + z.re));
        return cmk<T>(t, z.im / (T(2) * t));
    }
    T t = sqrt(T(0.5) * (r - z.re));
    return cmk<T>(fabs(z.im) / (T(2) * t), copysign(t, z.im));
}

}  // namespace vbx
