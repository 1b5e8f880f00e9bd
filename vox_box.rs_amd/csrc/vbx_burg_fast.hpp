// vbx_burg_fast.hpp (kernels; k_burg_fast.hip: dispatch; k_burg_fast_p*.hip: one instantiation per order) -- Burg LPC
// (LPC::lpc_praat_mut, src/spectrum.rs:101-146) in O(N P) + O(P^2) instead of O(N P) PER ORDER:
// the reflection coefficients from the frame's lag sums and its first and last P + 1 samples.
//
// The reference walks the forward / backward error arrays b2 / b1 once per order (:116-139).  Write a_i for the order-i
// prediction polynomial (a_i[0] = 1), f_i[n] = sum_k a_i[k] x[n-k] and b_i[n] = sum_k a_i[k] x[n-i+k]; the reference's
// numerator and denominator at order i + 1 are
//     num = sum_{n=i+1}^{N-1} f_i[n] b_i[n-1]       den = sum_{n=i+1}^{N-1} f_i[n]^2 + b_i[n-1]^2
// With A = [a_i, 0], B = reverse(A) (both of length i + 2) and the covariance matrix of the window [i+1, N)
//     R_i[p][q] = sum_{n=i+1}^{N-1} x[n-p] x[n-q],      num = B' R_i A,   den = A' R_i A + B' R_i B.
// R_i is never formed.  The kernel carries U = R_i A and V = R_i B and moves them to the next order in O(i):
//     A' = A - mu B  (extended by a zero),   B' = reverse(A'),   mu = 2 num / den
//     R_{i+1}[p][q]     = R_i[p][q] - e[p] e[q],   e[p] = x[i+1-p]        (the window loses its first sample)
//     R_{i+1}[p+1][q+1] = R_i[p][q] - t[p] t[q],   t[p] = x[N-1-p]        (the shifted window loses its last sample)
//     U'[p]   = U[p] - mu V[p] - e[p] (e . A')           p <= i + 1
//     V'[p+1] = V[p] - mu U[p] - t[p] (t . reverse(A'))  p <= i + 1
// and the one entry of each that is new,
//     U'[i+2] = sum_q A'[q] rho[i+2-q],   rho[d] = sum_{m=0}^{N-i-3} x[m] x[m+d]     (lag sums that stop early)
//     V'[0]   = sum_q B'[q] sig[q],       sig[q] = sum_{n=i+2}^{N-1} x[n] x[n-q]     (lag sums that start late)
// where rho and sig lose one product per order and gain the next full lag sum c[i+2].  So the frame is read ONCE
// (P + 1 lag sums: the few-lag autocorrelation of k_lpc.hip), and the recursion runs one frame per LANE.
//
// Two kernels.  burg_lags_kernel streams the frames (HBM- / issue-bound, 2-4 wavefronts per SIMD) and leaves 3 (P + 1)
// doubles per frame -- lag sums, first samples, last samples -- in a scratch tiled [64 frames][value][frame]; burg_recursion_kernel
// runs the recursion, 64 frames per wavefront.  Fully unrolled on registers the recursion wants ~290 of them: fused behind the
// streaming loop it cost that loop its occupancy (one wavefront per SIMD), and only the 16 lanes that own a frame of the
// wavefront's batch would run it.  The scratch (312 B per frame at order 12) is why the batch is cut into chunks of BF_CHUNK frames.
//
// Accuracy.  num and den are differences of terms of size c[0] |A|^2: the lag sums' own rounding (~eps c[0]) reaches mu
// amplified by kappa = c[0] |A|_1^2 / den.  On speech kappa eps is ~1e-12 (coefficients within ~1e-11 of the row's largest);
// the direct recursion of k_burg.hip is exact to ~eps.  The kernel therefore evaluates the bound itself: a frame whose
// coefficients could be off by more than BF_TARGET in the parity metric of tests/ (|d| <= 1e-6 max(|a_j|, 1e-6 max|a|)) --
// a badly conditioned frame, or one with a coefficient that happens to be tiny -- or whose denominator is not positive
// (the reference's Err(LPC), NaN input) is NOT written: its index goes to a list, and the direct kernel runs on the list
// (run_burg, vbx_api.hip).  About 1 % of speech frames at order 12 (0.25 % at order 8, 1.9 % at 16); a pure tone or a silent
// frame always.
#pragma once

#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

#include <type_traits>

namespace vbx {

#ifndef VBX_BF_FPW
#define VBX_BF_FPW 16
#endif
constexpr int BF_FPW = VBX_BF_FPW;         // frames per wavefront (burg_lags_kernel) ...
// ... but 8 for frames of up to 512 samples (8 samples per lane): config 4's lag kernel 0.47 -> 0.43 ms per 524,288 frames
// (lpc_burg at 512: 813 -> 907 M frames/s); at 1024 and 2048 samples 8 is slower (474 -> 451, 250 -> 230 M), at 1200 the same
template <int EPL> struct bf_fpw { static constexpr int value = (EPL == 8 && BF_FPW == 16) ? 8 : BF_FPW; };
#ifndef VBX_BF_CHUNK
#define VBX_BF_CHUNK 524288
#endif
constexpr long BF_CHUNK = VBX_BF_CHUNK;    // frames per pair of launches: 164 MB of scratch at order 12 (config 4, frames/s at
                                           // 131072 / 262144 / 524288 / 1048576 per chunk: 366 / 380 / 389 / 389 M)
// bound used by the guard: 64 kappa eps max|a| <= BF_TARGET * max(|a_j|, 1e-6 max|a|) for every j.  The observed error is
// at most 40 kappa eps max|a| (20,000 speech frames at 512 and at 1200 samples, 6,000 adversarial frames).
constexpr double BF_KAPPA_EPS = 64.0 * 2.220446049250313e-16;
constexpr double BF_TARGET = 5e-7;

// lane-per-frame recursion.  c[0..P]: lag sums; hd[k] = x[k], tl[k] = x[N-1-k], k = 0..P.  a[1..P]: the coefficients
// (reference sign: what lpc_praat_mut returns).  Returns true if the result is inside the guard.
template <int P>
__device__ __forceinline__ bool burg_from_lags(const double *c_, const double *hd_, const double *tl_, double (&a)[P + 1]) {
    // the inputs stay in memory (the wavefront's tile of the scratch: [value][64 frames], coalesced, L2-resident) and are
    // read where they are used, at compile-time offsets from one address
    auto c = [&](int k) { return c_[k * 64]; };
    auto hd = [&](int k) { return hd_[k * 64]; };
    auto tl = [&](int k) { return tl_[k * 64]; };
    double U[P + 2], V[P + 2], rho[P + 2], sig[P + 2];
#pragma unroll
    for (int k = 0; k <= P; k++) a[k] = 0.0;
#pragma unroll
    for (int k = 0; k < P + 2; k++) { U[k] = 0.0; V[k] = 0.0; rho[k] = 0.0; sig[k] = 0.0; }
    a[0] = 1.0;
    const double c0 = c(0), c1 = c(1);
    const double head = fma(-hd(0), hd(0), c0), tail = fma(-tl(0), tl(0), c0);
    U[0] = head; U[1] = c1;
    V[0] = c1; V[1] = tail;
    rho[0] = tail; rho[1] = c1;
    sig[0] = head; sig[1] = c1;
    bool ok = true;
    double kappa = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) {
        // A = a[0..i], A[i+1] = 0;  B[q] = A[i+1-q]
        asm volatile("" ::: "memory");                           // no hoisting of the later orders' loads: registers
        double num = 0.0, den = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 1; q <= i + 1; q++) { num = fma(a[i + 1 - q], U[q], num); den = fma(a[i + 1 - q], V[q], den); }
#pragma unroll
        for (int q = 0; q <= i; q++) { den = fma(a[q], U[q], den); s1 += fabs(a[q]); }
        ok = ok && (den > 0.0);                                  // false for NaN
        kappa = fmax(kappa, c0 * s1 * s1 * __builtin_amdgcn_rcp(den));
        const double mu = 2.0 * num / den;
        // a <- a - mu reverse(a) over [0, i + 1] (a[i+1] = 0 before: a[i+1] = -mu after), in place by pairs
#pragma unroll
        for (int k = 1; 2 * k <= i + 1; k++) {
            const int r = i + 1 - k;
            const double lo = a[k], hi = a[r];
            a[k] = fma(-mu, hi, lo);
            if (r != k) a[r] = fma(-mu, lo, hi);
        }
        a[i + 1] = -mu;
        if (i + 1 < P) {
            double fE = 0.0, bE = 0.0;
#pragma unroll
            for (int q = 0; q <= i + 1; q++) { fE = fma(hd(i + 1 - q), a[q], fE); bE = fma(tl(q), a[i + 1 - q], bE); }
            // U, V in place: V shifts up by one, so walk down
#pragma unroll
            for (int q = i + 1; q >= 0; q--) {
                const double u = U[q], v = V[q];
                V[q + 1] = fma(-tl(q), bE, fma(-mu, u, v));
                U[q] = fma(-hd(i + 1 - q), fE, fma(-mu, v, u));
            }
            const double xt = tl(i + 1), xh = hd(i + 1);
#pragma unroll
            for (int q = 0; q <= i + 1; q++) {
                rho[q] = fma(-xt, tl(i + 1 - q), rho[q]);
                sig[q] = fma(-xh, hd(i + 1 - q), sig[q]);
            }
            rho[i + 2] = c(i + 2);
            sig[i + 2] = c(i + 2);
            double un = 0.0, vn = 0.0;
#pragma unroll
            for (int q = 0; q <= i + 1; q++) { un = fma(a[q], rho[i + 2 - q], un); vn = fma(a[q], sig[i + 2 - q], vn); }
            U[i + 2] = un;
            V[0] = vn;
        }
    }
    double big = 0.0, small = __builtin_inf();
#pragma unroll
    for (int k = 1; k <= P; k++) { const double m = fabs(a[k]); big = fmax(big, m); small = fmin(small, m); }
    const double floor_j = fmax(small, 1e-6 * big);
    return ok && (BF_KAPPA_EPS * kappa * big <= BF_TARGET * floor_j);               // false for NaN
}

// EPL: samples per lane (frame_len <= 64 EPL).  P: the order.  TIN: double, or int16_t = 16-bit PCM widened in registers.
// One wavefront works through BF_FPW frames as autocorr_fewlags_kernel does (k_lpc.hip: EPL samples per lane in registers,
// neighbour samples by DPP wave shifts, one transposing reduction through LDS for all P + 1 lag sums).
// scratch: tiles of [3 (P + 1)][64] doubles, 64 consecutive items per tile.
template <int EPL, int P, typename TIN>
__global__ __launch_bounds__(64) void burg_lags_kernel(
    const TIN *__restrict__ x, long n_frames, int n, long stride, const double *__restrict__ window,
    const frame_map_t map, long item0, long items, double *__restrict__ scratch) {
    constexpr int FPW = bf_fpw<EPL>::value;
    constexpr int NL = P + 1;
    constexpr int TS = NL | 1;
    constexpr bool PCM = sizeof(TIN) == 2;
    constexpr int RAW = PCM ? EPL / 2 : EPL;         // registers of a frame in flight: packed pairs of PCM samples, or doubles
    static_assert(EPL % 2 == 0 && EPL >= 2, "pairs of samples per lane");
    __shared__ double TR[64 * TS];                   // per-frame transpose buffer [lane][lag]
    __shared__ double REC[3 * NL * FPW];             // [value][frame of the batch]: C, then HD, then TL
    const int lane = lane_id();
    const long i0 = item0 + (long)blockIdx.x * FPW;
    if (i0 >= item0 + items) return;
    const int nf = (int)((item0 + items - i0 < FPW) ? (item0 + items - i0) : FPW);

    double wreg[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        const int i = lane * EPL + e;
        wreg[e] = (i < n) ? ((window != nullptr) ? window[i] : 1.0) : 0.0;
    }
    const bool whole = (n % EPL == 0);               // no lane straddles the frame's end
    const bool mine = lane * EPL < n;
    using raw_t = typename std::conditional<PCM, uint32_t, double>::type;
    // 16-byte loads of doubles / 4-byte loads of PCM pairs where every row allows them
    const bool wide = whole && (PCM ? ((((uintptr_t)x) & 3) == 0 && (stride & 1) == 0)
                                    : ((((uintptr_t)x) & 15) == 0 && (stride & 1) == 0));
    auto load_frame = [&](int g, raw_t (&dst)[RAW]) {
        const long f = frame_map(map, i0 + g, n_frames);
        const TIN *xf = x + (f < 0 ? 0 : f) * stride + lane * EPL;
        const bool have = f >= 0 && mine;
        if constexpr (PCM) {
            if (wide) {
                const uint32_t *xv = reinterpret_cast<const uint32_t *>(xf);
#pragma unroll
                for (int e = 0; e < RAW; e++) dst[e] = have ? xv[e] : 0u;
            } else {
#pragma unroll
                for (int e = 0; e < RAW; e++) {
                    const int j = lane * EPL + 2 * e;
                    const uint32_t lo = (f >= 0 && j < n) ? (uint16_t)xf[2 * e] : 0u;
                    const uint32_t hi = (f >= 0 && j + 1 < n) ? (uint16_t)xf[2 * e + 1] : 0u;
                    dst[e] = lo | (hi << 16);
                }
            }
        } else {
            if (wide) {
                const double2 *xv = reinterpret_cast<const double2 *>(xf);
#pragma unroll
                for (int e = 0; e < EPL; e += 2) {
                    double2 v; v.x = 0.0; v.y = 0.0;
                    if (have) v = xv[e / 2];
                    dst[e] = v.x; dst[e + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPL; e++) dst[e] = (f >= 0 && lane * EPL + e < n) ? (double)xf[e] : 0.0;
            }
        }
    };
    raw_t cur[RAW], nxt[RAW], nx2[RAW];
    load_frame(0, cur);
    if (nf > 1) load_frame(1, nxt);
    const int red_lag = lane >> 2, red_part = lane & 3;

    for (int g = 0; g < nf; g++) {
        if (g + 2 < nf) load_frame(g + 2, nx2);
        double ext[EPL + NL - 1];
        if constexpr (PCM) {
#pragma unroll
            for (int e = 0; e < RAW; e++) {
                const int lo = (int)(int16_t)(cur[e] & 0xffffu), hi = (int)cur[e] >> 16;
                ext[2 * e] = pcm16_value(lo) * wreg[2 * e];
                ext[2 * e + 1] = pcm16_value(hi) * wreg[2 * e + 1];
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPL; e++) ext[e] = cur[e] * wreg[e];
        }
#pragma unroll
        for (int e = EPL; e < EPL + NL - 1; e++) ext[e] = from_next_lane(ext[e - EPL]);
        double part[NL];
#pragma unroll
        for (int lag = 0; lag < NL; lag++) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; e += 2) { s0 = fma(ext[e], ext[e + lag], s0); s1 = fma(ext[e + 1], ext[e + 1 + lag], s1); }
            part[lag] = s0 + s1;
        }
#pragma unroll
        for (int lag = 0; lag < NL; lag++) TR[lane * TS + lag] = part[lag];
        // the frame's first and last P + 1 samples
        if (lane * EPL <= P || (lane + 1) * EPL >= n - 1 - P) {
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int i = lane * EPL + e;
                if (i <= P) REC[(NL + i) * FPW + g] = ext[e];
                if (i < n && i >= n - 1 - P) REC[(2 * NL + (n - 1 - i)) * FPW + g] = ext[e];
            }
        }
        wave_sync();
#pragma unroll
        for (int lbase = 0; lbase < NL; lbase += 16) {   // 16 lags per sweep (4 lanes per lag)
            const int rl = lbase + red_lag;
            double tot = 0.0;
            if (rl < NL) {
#pragma unroll
                for (int t = 0; t < 16; t++) tot += TR[(red_part * 16 + t) * TS + rl];
            }
            tot += dpp_f64<DPP_QUAD_XOR1>(tot);
            tot += dpp_f64<0x4E>(tot);               // quad_perm [2,3,0,1]
            if (red_part == 0 && rl < NL) REC[rl * FPW + g] = tot;
        }
        wave_sync();
#pragma unroll
        for (int e = 0; e < RAW; e++) { cur[e] = nxt[e]; nxt[e] = nx2[e]; }
    }
    // the scratch is tiled [64 frames: one wavefront of the recursion][value][frame]: FPW consecutive doubles per value
    static_assert(64 % FPW == 0, "a batch stays inside one tile");
    const long c0 = i0 - item0;
    double *o = scratch + (c0 >> 6) * (3 * NL * 64) + (c0 & 63);
    for (int idx = lane; idx < 3 * NL * FPW; idx += 64) {
        const int v = idx / FPW, g = idx % FPW;
        if (g < nf) o[v * 64 + g] = REC[idx];
    }
}

// The same lag sums for LONGER frames (1281 .. 4096 samples), 16 samples per lane and SEGMENT of 1024 samples: with all of a
// lane's samples in registers (EPL = 32 above) the kernel wants 349 of them -- one wavefront per SIMD, and none at all beside
// a wavefront of the pipeline's analyze kernel (231 registers): at 1600 samples the formant chain's lag kernel took 11 ms,
// waiting for SIMDs to drain, where the 1200-sample one takes 1.2.  Here a frame is walked in segments; a lane's partial lag
// sums stay in registers across them, lane 63's look-ahead (the first P samples of the next segment) comes from a
// broadcast load, and the next segment's samples are in flight while this one's products are formed.  ~200 registers.
template <int P>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void burg_lags_seg_kernel(
    const double *__restrict__ x, long n_frames, int n, long stride, const double *__restrict__ window,
    const frame_map_t map, long item0, long items, double *__restrict__ scratch) {
    constexpr int EPL = 16, SEG = 64 * EPL, FPW = BF_FPW, NL = P + 1, TS = NL | 1;
    static_assert(P <= EPL, "the look-ahead stays inside the next lane's samples");
    __shared__ double TR[64 * TS];
    __shared__ double REC[3 * NL * FPW];
    const int lane = lane_id();
    const long i0 = item0 + (long)blockIdx.x * FPW;
    if (i0 >= item0 + items) return;
    const int nf = (int)((item0 + items - i0 < FPW) ? (item0 + items - i0) : FPW);
    const int nseg = (n + SEG - 1) / SEG;
    const int red_lag = lane >> 2, red_part = lane & 3;
    // windowed samples [base, base + EPL) of frame xf (zero past the frame)
    auto load_seg = [&](const double *xf, int base, double (&dst)[EPL]) {
        const bool al = ((((uintptr_t)xf) | ((uintptr_t)window)) & 15) == 0 && base + EPL <= n;
        if (al) {
#pragma unroll
            for (int e = 0; e < EPL; e += 2) {
                const double2 v = *reinterpret_cast<const double2 *>(xf + base + e);
                double2 w = double2{1.0, 1.0};
                if (window != nullptr) w = *reinterpret_cast<const double2 *>(window + base + e);
                dst[e] = v.x * w.x; dst[e + 1] = v.y * w.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int i = base + e;
                double v = 0.0;
                if (i < n) { v = xf[i]; if (window != nullptr) v *= window[i]; }
                dst[e] = v;
            }
        }
    };
    for (int g = 0; g < nf; g++) {
        const long f = frame_map(map, i0 + g, n_frames);
        const double *xf = x + (f < 0 ? 0 : f) * stride;
        const int nn = (f < 0) ? 0 : n;                       // an item outside the batch: zeros, nothing read
        double part[NL];
#pragma unroll
        for (int lag = 0; lag < NL; lag++) part[lag] = 0.0;
        double cur[EPL], nxt[EPL];
        if (nn > 0) load_seg(xf, lane * EPL, cur);
        else {
#pragma unroll
            for (int e = 0; e < EPL; e++) cur[e] = 0.0;
        }
        for (int sgm = 0; sgm < nseg; sgm++) {
            const int base = sgm * SEG + lane * EPL;
            const bool more = sgm + 1 < nseg && nn > 0;
            if (more) load_seg(xf, base + SEG, nxt);
            double ext[EPL + NL - 1];
#pragma unroll
            for (int e = 0; e < EPL; e++) ext[e] = cur[e];
            // look-ahead: the next lane's first P samples; lane 63's are the next segment's first P (one address for the wave)
#pragma unroll
            for (int k = 0; k < NL - 1; k++) {
                double ahead = from_next_lane(ext[k]);
                const int i = (sgm + 1) * SEG + k;
                double t = 0.0;
                if (more && i < n) { t = xf[i]; if (window != nullptr) t *= window[i]; }
                ext[EPL + k] = (lane == 63) ? t : ahead;
            }
#pragma unroll
            for (int lag = 0; lag < NL; lag++) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int e = 0; e < EPL; e += 2) { s0 = fma(ext[e], ext[e + lag], s0); s1 = fma(ext[e + 1], ext[e + 1 + lag], s1); }
                part[lag] += s0 + s1;
            }
            // the frame's first and last P + 1 samples
            if (nn > 0 && (base <= P || base + EPL >= n - 1 - P)) {
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    const int i = base + e;
                    if (i <= P) REC[(NL + i) * FPW + g] = ext[e];
                    if (i < n && i >= n - 1 - P) REC[(2 * NL + (n - 1 - i)) * FPW + g] = ext[e];
                }
            }
#pragma unroll
            for (int e = 0; e < EPL; e++) cur[e] = more ? nxt[e] : 0.0;
        }
#pragma unroll
        for (int lag = 0; lag < NL; lag++) TR[lane * TS + lag] = part[lag];
        wave_sync();
#pragma unroll
        for (int lbase = 0; lbase < NL; lbase += 16) {
            const int rl = lbase + red_lag;
            double tot = 0.0;
            if (rl < NL) {
#pragma unroll
                for (int t = 0; t < 16; t++) tot += TR[(red_part * 16 + t) * TS + rl];
            }
            tot += dpp_f64<DPP_QUAD_XOR1>(tot);
            tot += dpp_f64<0x4E>(tot);
            if (red_part == 0 && rl < NL) REC[rl * FPW + g] = tot;
        }
        wave_sync();
    }
    const long c0 = i0 - item0;
    double *o = scratch + (c0 >> 6) * (3 * NL * 64) + (c0 & 63);
    for (int idx = lane; idx < 3 * NL * FPW; idx += 64) {
        const int v = idx / FPW, g = idx % FPW;
        if (g < nf) o[v * 64 + g] = REC[idx];
    }
}

// lane <-> item: the recursion on the scratch's columns; the coefficient rows, the status, or the frame's index on the list
// Two wavefronts per SIMD (256 registers, some of the recursion's state spilled to private memory): unconstrained the
// compiler takes ~290, and then a wavefront of this kernel cannot start beside the pipeline's analyze kernel (two
// wavefronts of 232 registers per SIMD: one of them leaving frees 280) -- the whole formant chain waited for that kernel
// to drain (measured: 71 ms behind a 72 ms analyze launch).
template <int P>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void burg_recursion_kernel(
    const double *scratch /* no __restrict__: invariant loads would all be hoisted to the top */, long n_frames, const frame_map_t map, long item0, long items,
    double *__restrict__ out, int32_t *__restrict__ status, int32_t *__restrict__ list, int32_t *__restrict__ list_count) {
    constexpr int NL = P + 1;
    // the wavefront's tile of the scratch ([value][64 frames], contiguous) into LDS first: 3 (P + 1) coalesced loads in
    // flight at once.  Read from global memory where each order uses them (the memory barrier in burg_from_lags keeps the
    // loads there, for the registers' sake), every order waited a full memory latency: 99 us per 524,288 frames, of which
    // this takes two thirds.
    __shared__ double T[3 * NL * 64];
    const long col = (long)blockIdx.x * 64 + threadIdx.x;
    {
        const double *tile = scratch + (long)blockIdx.x * (3 * NL * 64);
#pragma unroll
        for (int v = 0; v < 3 * NL; v++) T[v * 64 + threadIdx.x] = tile[v * 64 + threadIdx.x];
    }
    wave_sync();
    if (col >= items) return;
    const long f = frame_map(map, item0 + col, n_frames);
    if (f < 0) return;
    double a[P + 1];
    const double *t = T + threadIdx.x;
    const bool trusted = burg_from_lags<P>(t, t + NL * 64, t + 2 * NL * 64, a);
    if (!trusted) {
        list[atomicAdd(list_count, 1)] = (int32_t)f;
        return;
    }
    double *o = out + f * (long)P;
#pragma unroll
    for (int k = 0; k < P; k++) o[k] = a[k + 1];
    if (status != nullptr) status[f] = 0;
}

// per-order launchers: explicit instantiations live in k_burg_fast_p<P>.hip (one translation unit per order keeps the build parallel)
template <int P, typename TIN>
void launch_burg_lags_p(hipStream_t s, const TIN *x, long F, int n, long stride, const double *window,
                        frame_map_t map, long i0, long m, double *scratch);
template <int P>
void launch_burg_recursion_p(hipStream_t s, const double *scratch, long F, frame_map_t map, long i0, long m,
                             double *out, int32_t *status, int32_t *list);

#define VBX_BURG_FAST_INSTANTIATE(P)                                                                                          \
    template <int PP, typename TIN>                                                                                           \
    void launch_burg_lags_p(hipStream_t s, const TIN *x, long F, int n, long stride, const double *window,                   \
                            frame_map_t map, long i0, long m, double *scratch) {                                              \
        const dim3 grid((unsigned)((m + BF_FPW - 1) / BF_FPW)), b(64);                                                        \
        const dim3 grid8((unsigned)((m + bf_fpw<8>::value - 1) / bf_fpw<8>::value));                                          \
        if (n <= 64 * 8) hipLaunchKernelGGL((burg_lags_kernel<8, PP, TIN>), grid8, b, 0, s, x, F, n, stride, window, map, i0, m, scratch);       \
        else if (n <= 64 * 16) hipLaunchKernelGGL((burg_lags_kernel<16, PP, TIN>), grid, b, 0, s, x, F, n, stride, window, map, i0, m, scratch); \
        else if (n <= 64 * 20) hipLaunchKernelGGL((burg_lags_kernel<20, PP, TIN>), grid, b, 0, s, x, F, n, stride, window, map, i0, m, scratch); \
        else if constexpr (std::is_same<TIN, double>::value) {                                                                \
            hipLaunchKernelGGL((burg_lags_seg_kernel<PP>), grid, b, 0, s, x, F, n, stride, window, map, i0, m, scratch);      \
        } else hipLaunchKernelGGL((burg_lags_kernel<32, PP, TIN>), grid, b, 0, s, x, F, n, stride, window, map, i0, m, scratch);                \
    }                                                                                                                         \
    template <int PP>                                                                                                         \
    void launch_burg_recursion_p(hipStream_t s, const double *scratch, long F, frame_map_t map, long i0, long m,              \
                                 double *out, int32_t *status, int32_t *list) {                                               \
        hipLaunchKernelGGL((burg_recursion_kernel<PP>), dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s, scratch, F, map, i0, m, \
                           out, status, list + 2, list);                                                                      \
    }                                                                                                                         \
    template void launch_burg_lags_p<P, double>(hipStream_t, const double *, long, int, long, const double *, frame_map_t, long, long, double *);   \
    template void launch_burg_lags_p<P, int16_t>(hipStream_t, const int16_t *, long, int, long, const double *, frame_map_t, long, long, double *); \
    template void launch_burg_recursion_p<P>(hipStream_t, const double *, long, frame_map_t, long, long, double *, int32_t *, int32_t *);

}  // namespace vbx
