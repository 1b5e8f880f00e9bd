// vbx_spectral_pow2.hpp -- the spectral pass of k_spectral.hip for the frame lengths the reference itself uses: 2048
// (examples/pitch_detection.rs:23, Windower::hanning(.., 2048, 1024)), 1024 (tests/lib.rs:56-57) and anything up to them.
//
// The same algebra as k_spectral.hip (see its header): ONE real FFT of the zero-padded windowed frame of length
// M = 2 Nc >= 2 n gives every lag sum S[lag] (|X|^2 -> second transform), S[0..12] feed the register Levinson, and -- when
// the frame fills the transform exactly (n == Nc) -- X[2k'] are the n-point DFT bins the mel filters read.  Nc is 1024,
// 2048 or 4096, the frame may be shorter (the padding is then longer than the frame; linear correlation needs M >= 2n - 1).
//
// One wavefront per frame.  Complex FFT of length Nc = 16 * 16 * R (R = 4, 8 or 16), decimation in frequency with
// n = 16R a + R b + c and k = ka + 16 kb + 256 kc:
//   stage 1  unit n' = R b + c (16R units, U = R/4 per lane: n' = lane + 64 u): 16-point DFT over a, times W_Nc^(n' ka)
//   stage 2  unit (ka, c):  16-point DFT over b, times W_16R^(c kb)
//   stage 3  lane q = ka + 16 kb (4 per lane): R-point DFT over c -> X[q + 256 kc], natural order
// between the stages the values change lanes through LDS, real and imaginary parts in two passes over one buffer that
// later holds the lag curve y.
#pragma once

#include "vbx_device.hpp"
#include "vbx_kernels.hpp"
#include "vbx_mfcc_tail.hpp"
#include "vbx_mfcc_interp.hpp"
#include "vbx_pitch_refine.hpp"
#include "vbx_spectral.hpp"

namespace vbx {

// y[lag] = (r[lag] * scale) / w_lag (src/periodic.rs:404-408) for the thread's NS pairs of lags (jj[s] = the pair's index; entries past
// nst are not stored), into the LDS curve or the split form's scratch row.  by_table: the divide by the window table's reciprocals
// (quotient_by_table, vbx_spectral.hpp), a few slots' window entries and reciprocals requested together and their quotients formed
// before the next slots' loads may start (a compiler barrier: all NS slots' pairs at once are registers the instances do not have).
template <int NS>
__device__ __forceinline__ void store_lag_curve(double *dst, const double (&r_e)[NS], const double (&r_o)[NS], const int (&jj)[NS], int nst,
                                                double scale, const double *__restrict__ lag_window, const double *__restrict__ lag_rcp,
                                                bool by_table) {
    if (by_table) {
        constexpr int LB = 4;
#pragma unroll
        for (int h = 0; h < (NS + LB - 1) / LB; h++) {
            double2 lwv[LB], rwv[LB];
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int s = LB * h + u;
                if (s >= NS) continue;
                const int i = 2 * jj[s];
                const int at = (i + 1 < nst) ? i : 0;
                lwv[u] = *reinterpret_cast<const double2 *>(lag_window + at);
                rwv[u] = *reinterpret_cast<const double2 *>(lag_rcp + at);
            }
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int s = LB * h + u;
                if (s >= NS) continue;
                const int i = 2 * jj[s];
                if (i + 1 < nst) {
                    double2 y;
                    y.x = quotient_by_table(r_e[s] * scale, lwv[u].x, rwv[u].x);
                    y.y = quotient_by_table(r_o[s] * scale, lwv[u].y, rwv[u].y);
                    *reinterpret_cast<double2 *>(dst + i) = y;
                } else if (i < nst) {                        // the last lag of an odd n
                    dst[i] = (r_e[s] * scale) / lag_window[i];
                }
            }
            asm volatile("" ::: "memory");
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int i = 2 * jj[s];
        if (i + 1 < nst) {
            const double2 lw = *reinterpret_cast<const double2 *>(lag_window + i);
            double2 y;
            y.x = (r_e[s] * scale) / lw.x;
            y.y = (r_o[s] * scale) / lw.y;
            *reinterpret_cast<double2 *>(dst + i) = y;
        } else if (i < nst) {                                // the last lag of an odd n
            dst[i] = (r_e[s] * scale) / lag_window[i];
        }
    }
}



#ifndef VBX_POW2_U1_WAVES
#define VBX_POW2_U1_WAVES 3                     // Nc = 1024, the fused analysis: three wavefronts per SIMD (168 registers, pinned twiddle batches)
#endif
#ifndef VBX_POW2_U2_WAVES
#define VBX_POW2_U2_WAVES 2
#endif

// U: 16-point units per THREAD and stage; W: wavefronts per frame (1, or 2 for the 4096-point transform: 64 complex values per
// lane of ONE wavefront are 512 registers + ~150 spilled at one wavefront per SIMD and three per CU; as two wavefronts each
// thread holds what a lane of the 2048-point kernel holds).  The transform's geometry depends on R = 4 U W alone.
template <int U, int W = 1>
struct pow2_geom {
    static constexpr int NT = 64 * W;                  // threads per frame
    static constexpr int R = 4 * U * W;                // radix of the last stage
    static constexpr int NC = 256 * R;                 // complex FFT length
    static constexpr int UNITS = 16 * R;               // 16-point DFTs per stage
    static constexpr int CW = NT / 16;                 // values of c per unit index u in stage 2 (4 per wavefront)
    static constexpr int TQ = 256 / NT;                // stage-3 columns q = tid + NT t per thread
    static constexpr int S1 = 16 * R + 2;              // exchange 1 row stride [ka][n']: lane (ka, c) reads 2 ka + c mod 32
    static constexpr int S2 = 256 + 16;                // exchange 2 row stride [c][ka + 16 kb]
    static constexpr int T1 = 0;                       // twiddle table (complex entries): T1[16R][16] = W_Nc^(n' ka)
    static constexpr int T2 = T1 + UNITS * 16;         //                                  T2[R][16]   = W_16R^(c kb)
    static constexpr int TM = T2 + R * 16;             //                                  WM[Nc/2+1]  = W_2Nc^m
    static constexpr int TAB = TM + NC / 2 + 1;
    static constexpr int TP = NC / (2 * NT) + 1;       // pairs (m, Nc - m), m = tid + NT t <= Nc / 2, per thread
    static constexpr int EX = (16 * S1 > R * S2) ? 16 * S1 : R * S2;   // doubles of the exchange buffer
    // round 5: the stage-2 twiddles T2[R][16] live in LDS behind the exchange buffer (the mel sums between the transforms never
    // reach past it: 2 (Nc / 2) + 64 <= EX), in a region only the refinement uses later: eight to sixteen dependent round trips
    // to the L1 / L2 per transform become LDS reads, and the kernel no longer carries the rows' 64-bit addresses
    static constexpr int T2_LDS_OFFSET = ((EX * 8 + 15) / 16) * 16;    // bytes
    static constexpr int T2_LDS_BYTES = R * 16 * 16;
    static_assert(NC + 64 <= EX, "the mel sums stay inside the exchange buffer");
};

// 16-point DFT in place, radix 4 x 4.  Input index a sits in slot a; output index k is left in slot dft16_slot(k).
__host__ __device__ constexpr int dft16_slot(int k) { return 4 * (k % 4) + k / 4; }

__device__ __forceinline__ void rot(double &r, double &i, double wr, double wi) {      // (r + i i) * (wr + i wi)
    const double a = r, b = i;
    r = fma(a, wr, -(b * wi));
    i = fma(a, wi, b * wr);
}

__device__ __forceinline__ void dft16(double (&re)[16], double (&im)[16]) {
    constexpr double C1 = 0.92387953251128675613;    // cos(pi/8)
    constexpr double S1 = 0.38268343236508977173;    // sin(pi/8)
    constexpr double R2 = 0.70710678118654752440;    // sqrt(1/2)
    // x[4 n1 + n2]: DFT over n1 for each n2 -> index k1 in slot 4 k1 + n2
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++)
        dft4(re[n2], im[n2], re[4 + n2], im[4 + n2], re[8 + n2], im[8 + n2], re[12 + n2], im[12 + n2]);
    // times W_16^(n2 k1)
    rot(re[5], im[5], C1, -S1);                      // k1 = 1: W^1, W^2, W^3
    { const double a = re[6], b = im[6]; re[6] = R2 * (a + b); im[6] = R2 * (b - a); }
    rot(re[7], im[7], S1, -C1);
    { const double a = re[9], b = im[9]; re[9] = R2 * (a + b); im[9] = R2 * (b - a); }   // k1 = 2: W^2, W^4, W^6
    { const double a = re[10], b = im[10]; re[10] = b; im[10] = -a; }
    { const double a = re[11], b = im[11]; re[11] = R2 * (b - a); im[11] = -(R2 * (a + b)); }
    rot(re[13], im[13], S1, -C1);                    // k1 = 3: W^3, W^6, W^9
    { const double a = re[14], b = im[14]; re[14] = R2 * (b - a); im[14] = -(R2 * (a + b)); }
    rot(re[15], im[15], -C1, S1);
    // DFT over n2 for each k1 -> index k2 in slot 4 k1 + k2, output k = k1 + 4 k2
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
        dft4(re[4 * k1], im[4 * k1], re[4 * k1 + 1], im[4 * k1 + 1], re[4 * k1 + 2], im[4 * k1 + 2], re[4 * k1 + 3], im[4 * k1 + 3]);
}

// R-point DFT of v[0..R) in natural order (R = 4: one butterfly; R = 8: radix 4 x 2; R = 16: radix 4 x 4)
template <int R>
__device__ __forceinline__ void dft_last(double (&vr)[R], double (&vi)[R], double (&xr)[R], double (&xi)[R]) {
    if constexpr (R == 4) {
        dft4(vr[0], vi[0], vr[1], vi[1], vr[2], vi[2], vr[3], vi[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) { xr[k] = vr[k]; xi[k] = vi[k]; }
    } else if constexpr (R == 16) {
        dft16(vr, vi);
#pragma unroll
        for (int k = 0; k < 16; k++) { xr[k] = vr[dft16_slot(k)]; xi[k] = vi[dft16_slot(k)]; }
    } else {
        constexpr double R2 = 0.70710678118654752440;
        // x[2 n1 + n2]: DFT over n1 for each n2 -> k1 in slot 2 k1 + n2
        dft4(vr[0], vi[0], vr[2], vi[2], vr[4], vi[4], vr[6], vi[6]);
        dft4(vr[1], vi[1], vr[3], vi[3], vr[5], vi[5], vr[7], vi[7]);
        // odd slots times W_8^k1
        { const double a = vr[3], b = vi[3]; vr[3] = R2 * (a + b); vi[3] = R2 * (b - a); }
        { const double a = vr[5], b = vi[5]; vr[5] = b; vi[5] = -a; }
        { const double a = vr[7], b = vi[7]; vr[7] = R2 * (b - a); vi[7] = -(R2 * (a + b)); }
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++) {
            xr[k1] = vr[2 * k1] + vr[2 * k1 + 1]; xi[k1] = vi[2 * k1] + vi[2 * k1 + 1];
            xr[k1 + 4] = vr[2 * k1] - vr[2 * k1 + 1]; xi[k1 + 4] = vi[2 * k1] - vi[2 * k1 + 1];
        }
    }
}

// thread index inside the frame's workgroup, and its barrier: one wavefront orders its own LDS traffic, two need s_barrier
template <int W> __device__ __forceinline__ int pow2_tid() { return W == 1 ? lane_id() : (int)threadIdx.x; }
template <int W> __device__ __forceinline__ void pow2_sync() { if constexpr (W == 1) wave_sync(); else __syncthreads(); }

// Complex FFT of length Nc.  In: thread i holds z[16R a + i + NT u] in (re[u][a], im[u][a]).  Out: thread i holds
// X[i + NT t + 256 kc] in (xr[t][kc], xi[t][kc]).  ex: LDS exchange buffer (pow2_geom<U, W>::EX doubles).
// The sixteen twiddle products of a unit in batches of FOUR, each batch finished before the next one's loads may start (the
// products pinned by an empty asm, the loads fenced by a compiler memory barrier): left alone the compiler requests every
// twiddle of the stage right after the 16-point DFTs and spills their results to make room (k_spectral.hip, twiddle_tight).
// (batches of FOUR for the two-wavefront form of Nc = 4096; batches of TWO for one wavefront per frame -- with them and the
// per-call table pointer below the fused Nc = 2048 kernel spills nothing (round 4: 9 registers = 2.5 KB of scratch per frame,
// the "1.31x traffic" of the round-4 review), the Nc = 1024 kernels 0-2 registers instead of 3-7)
template <int B>
__device__ __forceinline__ void twiddle_tight16(double (&re)[16], double (&im)[16], const double2 *tw_row) {
#pragma unroll
    for (int h = 0; h < 16 / B; h++) {
        double2 w[B];
#pragma unroll
        for (int k = 0; k < B; k++) w[k] = tw_row[B * h + k];
#pragma unroll
        for (int k = 0; k < B; k++) {
            if (B * h + k == 0) continue;
            const int s = dft16_slot(B * h + k);
            rot(re[s], im[s], w[k].x, w[k].y);
            asm volatile("" : "+v"(re[s]), "+v"(im[s]));
        }
        asm volatile("" ::: "memory");
    }
}
// TIGHT: twiddle_tight16 (the instances that are short of registers: three wavefronts per SIMD at Nc = 1024, and Nc = 2048)
template <int U, int W = 1, bool TIGHT = false>
__device__ __forceinline__ void fft_pow2(double (&re)[U][16], double (&im)[U][16], double (&xr)[4 / W][4 * U * W],
                                         double (&xi)[4 / W][4 * U * W], double *ex, const double2 *tab, const double2 *t2 = nullptr /* T2 in LDS, or from the table */) {
    using G = pow2_geom<U, W>;
    constexpr int R = G::R, NT = G::NT, CW = G::CW, TQ = G::TQ;
    const int tid = pow2_tid<W>();
    // the table pointer is made opaque per call: the two transforms of a kernel would otherwise SHARE the per-lane twiddle
    // row addresses -- computed in the first, kept (spilled: five 64-bit values per lane at Nc = 2048, 2.5 KB of scratch
    // written and read back per frame) for the second.  Recomputing them costs a few integer instructions.
    if constexpr (W == 1) asm volatile("" : "+s"(tab));
    constexpr int TB = (W == 1) ? 2 : 4;
    // stage 1
#pragma unroll
    for (int u = 0; u < U; u++) {
        dft16(re[u], im[u]);
        const double2 *tw = tab + G::T1 + (tid + NT * u) * 16;
        if constexpr (TIGHT) { twiddle_tight16<TB>(re[u], im[u], tw); continue; }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double2 w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = tw[8 * h + k];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (8 * h + k == 0) continue;
                const int s = dft16_slot(8 * h + k);
                rot(re[u][s], im[u][s], w[k].x, w[k].y);
            }
        }
    }
    // exchange 1: [ka][n'] -> unit (ka2, c2) reads n' = R b + c2
    const int ka2 = tid & 15, c2 = tid >> 4;                // unit e = tid + NT u: (ka2, c2 + CW u)
    double br[U][16], bi[U][16];
    pow2_sync<W>();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int k = 0; k < 16; k++) ex[k * G::S1 + tid + NT * u] = re[u][dft16_slot(k)];
    pow2_sync<W>();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int b = 0; b < 16; b++) br[u][b] = ex[ka2 * G::S1 + R * b + c2 + CW * u];
    pow2_sync<W>();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int k = 0; k < 16; k++) ex[k * G::S1 + tid + NT * u] = im[u][dft16_slot(k)];
    pow2_sync<W>();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int b = 0; b < 16; b++) bi[u][b] = ex[ka2 * G::S1 + R * b + c2 + CW * u];
    // stage 2
#pragma unroll
    for (int u = 0; u < U; u++) {
        dft16(br[u], bi[u]);
        const double2 *tw = (t2 != nullptr ? t2 : tab + G::T2) + (c2 + CW * u) * 16;
        if constexpr (TIGHT) { twiddle_tight16<TB>(br[u], bi[u], tw); continue; }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double2 w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = tw[8 * h + k];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (8 * h + k == 0) continue;
                const int s = dft16_slot(8 * h + k);
                rot(br[u][s], bi[u][s], w[k].x, w[k].y);
            }
        }
    }
    // exchange 2: [c][ka + 16 kb] -> thread i reads q = i + NT t for every c
    double vr[TQ][R], vi[TQ][R];
    pow2_sync<W>();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int k = 0; k < 16; k++) ex[(c2 + CW * u) * G::S2 + ka2 + 16 * k] = br[u][dft16_slot(k)];
    pow2_sync<W>();
#pragma unroll
    for (int t = 0; t < TQ; t++)
#pragma unroll
        for (int c = 0; c < R; c++) vr[t][c] = ex[c * G::S2 + tid + NT * t];
    pow2_sync<W>();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int k = 0; k < 16; k++) ex[(c2 + CW * u) * G::S2 + ka2 + 16 * k] = bi[u][dft16_slot(k)];
    pow2_sync<W>();
#pragma unroll
    for (int t = 0; t < TQ; t++)
#pragma unroll
        for (int c = 0; c < R; c++) vi[t][c] = ex[c * G::S2 + tid + NT * t];
    pow2_sync<W>();
    // stage 3
#pragma unroll
    for (int t = 0; t < TQ; t++) dft_last<R>(vr[t], vi[t], xr[t], xi[t]);
}

// Nc = 4096 (frames of 2049..4096 samples): <U = 2, W = 2>, two wavefronts per frame (round 4).  As ONE wavefront (U = 4: 64
// complex values per lane) the kernel took 512 registers + ~150 spilled at one wavefront per SIMD, three frames = three
// wavefronts per CU (the frame state is 46 KB of LDS): 5.7 M frames/s at 4096 / 2048.  The second wavefront leaves when the lag
// curve is in LDS; the first runs the refinement as in every other kernel.
// U = 1 (Nc = 1024): three wavefronts per SIMD since the end of round 4 (168 registers with the twiddle products in pinned
// batches, twiddle_tight16; ~210 registers at two).  U = 2 with W = 1: two wavefronts per SIMD.  U = 2 (Nc = 2048: 32 complex values per lane, 17
// spectrum pairs) needs ~290: at 256 a dozen to fifty of them spill, and the frame state (23 KB of LDS) admits six frames
// per CU.  Measured at n = 2048: 25.2 M frames/s against 18.6 M with one wavefront per SIMD and no spills.
// FULL: the frame fills the transform (n == Nc, the bounds tests fold away); otherwise n < Nc (MFCC joins when n divides M, or -- the
// INTERP modes -- by interpolated bins).
// MODE (vbx_spectral.hpp): SP_ANALYZE the fused analysis; SP_MFCC_ONLY MFCC::mfcc alone (the forward transform and the mel / DCT
// tail only); SP_AC_ONLY Autocorrelate::autocorrelate alone (both transforms, the fold seed, the lag sums stored).
template <int U, bool LPC, bool MFCC, bool FULL, int MODE = SP_ANALYZE, int W = 1>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu((U == 1 && W == 1 && sp_is_analyze(MODE)) ? VBX_POW2_U1_WAVES : U == 4 ? 1 : (U == 2 && !sp_is_mfcc_only(MODE)) ? VBX_POW2_U2_WAVES : 2,
                                                                     (U == 1 && W == 1 && sp_is_analyze(MODE)) ? VBX_POW2_U1_WAVES : U == 4 ? (!sp_is_mfcc_only(MODE) ? 1 : 2) : !sp_is_mfcc_only(MODE) ? 2 : 4)))
void analyze_pow2_kernel(const spectral_args_t a) {
    // pinned twiddle batches where registers are short: Nc = 1024 at three wavefronts per SIMD, Nc = 2048 at two
    constexpr bool INTERP = sp_is_interp(MODE);              // MFCC's bins lie between the transform's (mfcc_interp_t, vbx_kernels.hpp)
    constexpr bool SPLIT = sp_is_split(MODE);                // the lag curve goes to HBM, refine_curve_kernel takes it from there
    static_assert(!INTERP || (MFCC && !FULL), "interpolated bins: a padded frame's MFCC");
    constexpr bool POW2_TIGHT = sp_is_analyze(MODE) && W == 1 && ((U == 1 && VBX_POW2_U1_WAVES >= 3) || U == 2);
    static_assert((MODE != SP_MFCC_ONLY && MODE != SP_MFCC_HALF) || (MFCC && FULL && !LPC), "the MFCC-only forms need the full frame and have no lag sums");
    static_assert(MODE != SP_MFCC_ONLY_INTERP || !LPC, "the MFCC-only forms have no lag sums");
    static_assert(MODE != SP_AC_ONLY || (!MFCC && !LPC), "the autocorrelation-only form");
    constexpr bool PITCH = !sp_is_mfcc_only(MODE);           // the second transform runs
    constexpr bool HALF = MODE == SP_MFCC_HALF;              // the frame has 2 Nc samples: every slot of the transform is data
    using G = pow2_geom<U, W>;
    constexpr int R = G::R, NC = G::NC, TP = G::TP, NT = G::NT, TQ = G::TQ;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double bcast[4 + W];                          // W > 1: values one wavefront hands the other (x0, S[0], the row maximum)
    const long fb = xcd_item(blockIdx.x, a.n_batch);        // (a launch covers frames [f0, f0 + n_batch): all of them unless the split form walks the batch in pieces)
    if (fb >= a.n_batch) return;
    const long f = a.f0 + fb;
    const int lane = lane_id();
    const int tid = pow2_tid<W>();                           // index inside the frame's workgroup (== lane for one wavefront)
    const int wave = W == 1 ? 0 : (int)(threadIdx.x >> 6);
    const int n = HALF ? 2 * NC : FULL ? NC : a.n;           // frame length, <= NC (SP_MFCC_HALF: 2 NC)
    double *ex = smem;                                       // exchange buffer, later the lag curve y
    const double *xf = a.frames + f * a.stride;
    double2 *t2 = reinterpret_cast<double2 *>(reinterpret_cast<char *>(smem) + G::T2_LDS_OFFSET);

    // ---- load: z[16R a + n'] = (xw[32R a + 2 n'], xw[.. + 1]), a < 8 (the rest is the zero padding), 0 past the frame ----
    double re[U][16], im[U][16];
    {
        const bool al = ((((uintptr_t)xf) | ((uintptr_t)a.window)) & 15) == 0;      // uniform
#pragma unroll
        for (int u = 0; u < U; u++) {
            constexpr int NQ = HALF ? 16 : 8;               // slots that hold samples (the rest is the zero padding)
            double2 xv[NQ], wv[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int i = 32 * R * q + 2 * (tid + NT * u);
                xv[q] = double2{0.0, 0.0}; wv[q] = double2{1.0, 1.0};
                if (al && i + 1 < n) {
                    xv[q] = *reinterpret_cast<const double2 *>(xf + i);
                    if (a.window != nullptr) wv[q] = *reinterpret_cast<const double2 *>(a.window + i);
                } else {
                    if (i < n) { xv[q].x = xf[i]; if (a.window != nullptr) wv[q].x = a.window[i]; }
                    if (i + 1 < n) { xv[q].y = xf[i + 1]; if (a.window != nullptr) wv[q].y = a.window[i + 1]; }
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                re[u][q] = (a.window != nullptr) ? xv[q].x * wv[q].x : xv[q].x;
                im[u][q] = (a.window != nullptr) ? xv[q].y * wv[q].y : xv[q].y;
            }
#pragma unroll
            for (int q = NQ; q < 16; q++) { re[u][q] = 0.0; im[u][q] = 0.0; }
        }
    }
    double x0 = readlane_f64(re[0][0], 0);                  // x_w[0], for the fold seed (Q1)
    if constexpr (W == 1) {                                  // (two wavefronts per frame: measured slower, the table stays in memory)
#pragma unroll
        for (int i = tid; i < R * 16; i += NT) t2[i] = a.tab[G::T2 + i];  // ordered before its first use by the first exchange's barriers
    }
    if constexpr (W > 1) {
        if (tid == 0) bcast[0] = x0;
        __syncthreads();
        x0 = bcast[0];
    }

    // ---- forward transform of the packed frame ----
    double xr[TQ][R], xi[TQ][R];
    fft_pow2<U, W, POW2_TIGHT>(re, im, xr, xi, ex, a.tab, W == 1 ? t2 : nullptr);

    // ---- exchange 3: natural order, then each lane takes the pairs (m, Nc - m), m = lane + 64 t <= Nc / 2 ----
    double ar[TP], ai[TP], br[TP], bi[TP];
#pragma unroll
    for (int t = 0; t < TQ; t++)
#pragma unroll
        for (int kc = 0; kc < R; kc++) ex[tid + NT * t + 256 * kc] = xr[t][kc];
    pow2_sync<W>();
#pragma unroll
    for (int t = 0; t < TP; t++) {
        const int m = tid + NT * t;
        const bool ok = m <= NC / 2;
        ar[t] = ok ? ex[m] : 0.0;
        br[t] = ok ? ex[(m == 0) ? 0 : NC - m] : 0.0;
    }
    pow2_sync<W>();
#pragma unroll
    for (int t = 0; t < TQ; t++)
#pragma unroll
        for (int kc = 0; kc < R; kc++) ex[tid + NT * t + 256 * kc] = xi[t][kc];
    pow2_sync<W>();
#pragma unroll
    for (int t = 0; t < TP; t++) {
        const int m = tid + NT * t;
        const bool ok = m <= NC / 2;
        ai[t] = ok ? ex[m] : 0.0;
        bi[t] = ok ? ex[(m == 0) ? 0 : NC - m] : 0.0;
    }
    pow2_sync<W>();

    // ---- spectrum of the real sequence, powers, the inverse transform's input (k_spectral.hip: same formulas) ----
    const int b_lo = MFCC ? a.bins[0] : 0;
    double pk[TP], pn[TP];                                   // P[m], P[Nc - m]
    double2 *zc = reinterpret_cast<double2 *>(ex);           // INTERP: Z[j - jmin] = X_M[j] e^{2 pi i j c / M}, Z[-j] = conj Z[j]
    double2 rot_m = double2{1.0, 0.0}, rot_step = double2{1.0, 0.0};
    if constexpr (INTERP) { rot_m = reinterpret_cast<const double2 *>(a.ip.rot)[tid]; rot_step = reinterpret_cast<const double2 *>(a.ip.rot)[NT]; }
#pragma unroll
    for (int t = 0; t < TP; t++) {
        const int m = tid + NT * t;
        const double2 w = a.tab[G::TM + ((m <= NC / 2) ? m : 0)];
        const double er = 0.5 * (ar[t] + br[t]), ei = 0.5 * (ai[t] - bi[t]);
        const double o_r = 0.5 * (ai[t] + bi[t]), o_i = -0.5 * (ar[t] - br[t]);
        const double tr = fma(w.x, o_r, -(w.y * o_i)), ti = fma(w.x, o_i, w.y * o_r);
        const double pr = er + tr, pi = ei + ti, qr = er - tr, qi = ei - ti;
        pk[t] = fma(pr, pr, pi * pi);
        pn[t] = fma(qr, qr, qi * qi);
        if constexpr (INTERP) {                              // (every thread is past exchange 3's last read: the buffer is free)
            asm volatile("" : "+v"(pk[t]), "+v"(pn[t]));     // the powers NOW: two values wait for exchange 4, not the four they are made of
            mfcc_interp_stage(zc, a.ip, m, pr, pi, rot_m, rot_step, t == 0);
        }
    }

    // ---- MFCC::mfcc at a length that does not divide the transform (k_spectral.hip: the same block; thread i: bins b_lo + i + NT u) ----
    if constexpr (INTERP) {
        pow2_sync<W>();
        const int nbp = (a.nb + 1) & ~1;
        double *pu = ex + a.ip.pu_off, *pd = pu + nbp, *en = pd + nbp;
        const double2 *cf = reinterpret_cast<const double2 *>(a.ip.coef) + tid;
        const int HT = a.ip.taps >> 1;                       // 12, 16 or 20 pairs of taps (the host's choice for M / n)
        for (int u = 0; u * NT < a.nb; u++) {
            const int b = tid + NT * u;
            const double2 *zp = zc + a.ip.j0[u * NT + tid];
            const double2 sl = *reinterpret_cast<const double2 *>(a.slopes + 2 * ((b < a.nb) ? b : 0));
            double vr, vi;
            mfcc_interp_bin(HT, cf + (u * HT) * NT, NT, zp, vr, vi);
            const double pw = fma(vr, vr, vi * vi);
            if (b < a.nb) {
                pu[b] = fabs(pw) * sl.x;                     // norm_sqr * multiplier (src/spectrum.rs:426-428)
                pd[b] = fabs(sqrt(pw)) * sl.y;               // norm * multiplier (:432-434)
            }
        }
        pow2_sync<W>();
        double2 t2v[(R * 16 + NT - 1) / NT];                 // the products may lie over the stage-2 twiddles (W == 1): requested now, put back after the tail
        if constexpr (W == 1 && PITCH) {
#pragma unroll
            for (int i = 0; i < (R * 16 + NT - 1) / NT; i++) t2v[i] = a.tab[G::T2 + ((tid + NT * i < R * 16) ? tid + NT * i : 0)];
        }
        if (wave == 0) {
            if (a.num_coeffs <= 16) mfcc_tail_q(pu, pd, en, a.bins, a.dct, a.num_coeffs, b_lo, lane, a.out_mfcc + f * a.mfcc_ld);
            else mfcc_tail_m(pu, pd, en, a.bins, a.dct, a.num_coeffs, b_lo, lane, a.out_mfcc + f * a.mfcc_ld);
            if (a.mfcc_status != nullptr && lane == 0) a.mfcc_status[f] = 0;
        }
        pow2_sync<W>();
        if constexpr (W == 1 && PITCH) {
#pragma unroll
            for (int i = 0; i < (R * 16 + NT - 1) / NT; i++) if (tid + NT * i < R * 16) t2[tid + NT * i] = t2v[i];
        }
    }

    if constexpr (PITCH) {
        // ---- exchange 4: G in natural order -> stage-1 layout of the second transform ----
    #pragma unroll
        for (int t = 0; t < TP; t++) {
            const int m = tid + NT * t;
            if (m <= NC / 2) {
                const double2 w = a.tab[G::TM + m];
                const double sm = pk[t] + pn[t], d = pk[t] - pn[t];
                ex[m] = fma(d, w.y, sm);
                if (m >= 1 && m < NC / 2) ex[NC - m] = fma(-d, w.y, sm);
            }
        }
        pow2_sync<W>();
    #pragma unroll
        for (int u = 0; u < U; u++)
    #pragma unroll
            for (int q = 0; q < 16; q++) re[u][q] = ex[16 * R * q + tid + NT * u];
        pow2_sync<W>();
    #pragma unroll
        for (int t = 0; t < TP; t++) {
            const int m = tid + NT * t;
            if (m <= NC / 2) {
                const double2 w = a.tab[G::TM + m];
                const double gi = -((pk[t] - pn[t]) * w.x);
                ex[m] = gi;
                if (m >= 1 && m < NC / 2) ex[NC - m] = gi;
            }
        }
        pow2_sync<W>();
    #pragma unroll
        for (int u = 0; u < U; u++)
    #pragma unroll
            for (int q = 0; q < 16; q++) im[u][q] = ex[16 * R * q + tid + NT * u];
        pow2_sync<W>();
    }

    // ---- MFCC::mfcc from the powers: X_n[k'] = X_M[q k'], q = M / n -- 2 for a frame of Nc samples, 1 for one of 2 Nc (HALF),
    //      4, 8, .. for a shorter frame whose length divides M (512 in the 1024 plan) ----
    if constexpr (MFCC && !INTERP) {
        const int nbp = (a.nb + 1) & ~1;
        const int qm = HALF ? 0 : FULL ? 1 : a.mfcc_q - 1;    // q is a power of two: m % q == 0  <=>  (m & (q - 1)) == 0
        const int qs = HALF ? 0 : FULL ? 1 : 31 - __builtin_clz((unsigned)a.mfcc_q);
        const int half = HALF ? NC : FULL ? NC / 2 : n / 2;
        double *pu = ex, *pd = ex + nbp, *en = ex + 2 * nbp; // the exchange buffer is free between the two transforms
        // (k_spectral.hip: when no mirrored bin can be one of the filters' -- they end below a quarter of the sampling rate --
        // only P[m] has bins, and the slope pairs of three slots are requested together, without a condition)
        const bool two_sided = (half - ((NC / 2) >> qs)) - b_lo < a.nb;
        if (!two_sided) {
            constexpr int MB1 = 3;
#pragma unroll
            for (int h = 0; h < (TP + MB1 - 1) / MB1; h++) {
                double2 s1[MB1];
                int c1[MB1];
#pragma unroll
                for (int u = 0; u < MB1; u++) {
                    const int t = MB1 * h + u, m = tid + NT * t;
                    if (t >= TP) { c1[u] = -1; continue; }
                    const int b1 = (m >> qs) - b_lo;
                    c1[u] = (m <= NC / 2 && (m & qm) == 0 && b1 >= 0 && b1 < a.nb) ? b1 : -1;
                    s1[u] = *reinterpret_cast<const double2 *>(a.slopes + 2 * (c1[u] < 0 ? 0 : c1[u]));
                }
#pragma unroll
                for (int u = 0; u < MB1; u++) {
                    const int t = MB1 * h + u;
                    if (t < TP && c1[u] >= 0) {
                        pu[c1[u]] = fabs(pk[t]) * s1[u].x;   // norm_sqr * multiplier (src/spectrum.rs:426-428)
                        pd[c1[u]] = fabs(sqrt(pk[t])) * s1[u].y;   // norm * multiplier (:432-434)
                    }
                }
            }
        } else
#pragma unroll
        for (int t = 0; t < TP; t++) {
            const int m = tid + NT * t;
            if (m <= NC / 2 && (m & qm) == 0) {
                const int b1 = (m >> qs) - b_lo, b2 = (half - (m >> qs)) - b_lo;
                if (b1 >= 0 && b1 < a.nb) {
                    const double2 sl = *reinterpret_cast<const double2 *>(a.slopes + 2 * b1);
                    pu[b1] = fabs(pk[t]) * sl.x;             // norm_sqr * multiplier (src/spectrum.rs:426-428)
                    pd[b1] = fabs(sqrt(pk[t])) * sl.y;       // norm * multiplier (:432-434)
                }
                if (b2 >= 0 && b2 < a.nb && b2 != b1) {
                    const double2 sl = *reinterpret_cast<const double2 *>(a.slopes + 2 * b2);
                    pu[b2] = fabs(pn[t]) * sl.x;
                    pd[b2] = fabs(sqrt(pn[t])) * sl.y;
                }
            }
        }
        pow2_sync<W>();
        if (wave == 0) {
            if (a.num_coeffs <= 16) mfcc_tail_q(pu, pd, en, a.bins, a.dct, a.num_coeffs, b_lo, lane, a.out_mfcc + f * a.mfcc_ld);
            else mfcc_tail_m(pu, pd, en, a.bins, a.dct, a.num_coeffs, b_lo, lane, a.out_mfcc + f * a.mfcc_ld);
            if (a.mfcc_status != nullptr && lane == 0) a.mfcc_status[f] = 0;
        }
        pow2_sync<W>();
    }

    if constexpr (!PITCH) return;

    // ---- second transform: Y = FFT(G);  S[2j] = Re Y[j] / M, S[2j+1] = -Im Y[j] / M, j = lane + 64 t + 256 kc < Nc / 2 ----
    fft_pow2<U, W, POW2_TIGHT>(re, im, xr, xi, ex, a.tab, W == 1 ? t2 : nullptr);

    constexpr int NS = TQ * (R / 2);                         // slots per thread: t < TQ, kc < R / 2
    constexpr double INV_M = 1.0 / (double)(2 * NC);
    double r_e[NS], r_o[NS];
    int jj[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int t = s % TQ, kc = s / TQ;
        jj[s] = tid + NT * t + 256 * kc;
        r_e[s] = xr[t][kc] * INV_M;
        r_o[s] = -(xi[t][kc] * INV_M);
    }
    double s0 = readlane_f64(r_e[0], 0);                     // S[0], the scale of the transform's rounding error
    if constexpr (W > 1) {
        if (tid == 0) bcast[1] = s0;
        __syncthreads();
        s0 = bcast[1];
    }
    if (x0 != 0.0) {                                         // rectangular frames: the fold seed differs from S (uniform branch)
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int i = 2 * jj[s];
            if (i < n) {
                const double xe = (a.window != nullptr) ? xf[i] * a.window[i] : xf[i];
                r_e[s] = (r_e[s] - x0 * xe) + x0;
            }
            if (i + 1 < n) {
                const double xo = (a.window != nullptr) ? xf[i + 1] * a.window[i + 1] : xf[i + 1];
                r_o[s] = (r_o[s] - x0 * xo) + x0;
            }
        }
    }
    if constexpr (MODE == SP_AC_ONLY) {                      // autocorrelate(n_lags): the lag sums and nothing else
        double *row = a.out_r + f * (long)a.n_lags;
        const bool al = ((((uintptr_t)a.out_r) & 15) == 0) && (a.n_lags & 1) == 0;      // uniform: every row 16-byte aligned
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int i = 2 * jj[s];
            if (i + 1 < a.n_lags && al) *reinterpret_cast<double2 *>(row + i) = double2{r_e[s], r_o[s]};
            else {
                if (i < a.n_lags) row[i] = r_e[s];
                if (i + 1 < a.n_lags) row[i + 1] = r_o[s];
            }
        }
        return;
    }
    if (LPC && wave == 0) {                                               // the raw autocorrelation r[0..12] into the frame's LPC row: lane l holds r[2l], r[2l + 1];
        // levinson_rows_kernel_t makes it LPC::lpc(12) afterwards, one row per lane (vbx_spectral.hpp)
        static_assert(SP_LPC_P == 12, "seven lanes hold r[0..12]");
        double *row = a.out_lpc + f * a.lpc_ld;
        if (lane < 7) { row[2 * lane] = r_e[0]; if (lane < 6) row[2 * lane + 1] = r_o[0]; }
    }
    double amax = -1.0;                                      // max_amplitude over the n lags (Q2; NaN never wins)
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int i = 2 * jj[s];
        const double ae = fabs(r_e[s]), ao = fabs(r_o[s]);
        if (i < n) amax = (ae > amax) ? ae : amax;
        if (i + 1 < n) amax = (ao > amax) ? ao : amax;
    }
    amax = wave_max(amax);
    if constexpr (W > 1) {                                   // NaN never wins in either wavefront, nor here
        if (lane == 0) bcast[4 + wave] = amax;
        __syncthreads();
        double m = bcast[4];
#pragma unroll
        for (int w = 1; w < W; w++) m = (bcast[4 + w] > m) ? bcast[4 + w] : m;
        amax = m;
    }
    const double scale = 1.0 / amax;                         // normalize (:404), then / lag window (:406-408)
    double *ys = smem;
    const int nst = (a.pp.ncurve > 0) ? a.pp.ncurve : n;     // lags the refinement can read (pitch_curve_entries; even when < n)
    // uniform: the lag-window divide by the table's reciprocals (quotient_by_table, vbx_spectral.hpp) unless the scale is not a normal finite number
    const bool by_table = (a.pcm & SP_FLAG_LAG_RCP) != 0 && fabs(scale) < 1e290 && fabs(scale) > 1e-290;
    const double *lag_rcp = a.lag_window + lag_rcp_offset(n);
    if constexpr (SPLIT) {                                   // the same values, to the frame's scratch row
        double *row = a.curve + fb * a.curve_ld;
        store_lag_curve<NS>(row, r_e, r_o, jj, nst, scale, a.lag_window, lag_rcp, by_table);
        if (tid < Y_PAD + (nst & 1)) row[nst + tid] = 0.0;   // (an odd n: one more, its row is read in pairs)
        if (tid == 0) a.curve_tol[fb] = SP_UNC_EPS * fabs(s0) * scale;
#ifndef VBX_EXP_NO_EXACT_TAIL
        if (!FULL && nst == n) {                             // an odd n keeps every lag: the last sixteen exactly, as in the fused kernel
            if constexpr (W > 1) { __syncthreads(); if (wave != 0) return; }
            spectral_exact_tail(row, n, xf, a.window, a.lag_window, x0, scale, lane);
        }
#endif
        return;
    }
    pow2_sync<W>();                                          // every thread is done with the exchange buffer
    store_lag_curve<NS>(ys, r_e, r_o, jj, nst, scale, a.lag_window, lag_rcp, by_table);
    if (tid < Y_PAD) ys[nst + tid] = 0.0;
    if constexpr (W > 1) {
        __syncthreads();                                     // the curve is complete in LDS: the refinement is one wavefront's work
        if (wave != 0) return;
    }
#ifndef VBX_EXP_NO_EXACT_TAIL
    if (!FULL && nst == n) spectral_exact_tail(ys, n, xf, a.window, a.lag_window, x0, scale, lane);
#endif
    wave_sync();
    const double unc_tol = SP_UNC_EPS * fabs(s0) * scale;
    double2 *full = a.pp.full_off ? reinterpret_cast<double2 *>(reinterpret_cast<char *>(smem) + a.pp.full_off) : nullptr;
    if (!pitch_refine_store(ys, n, a.pp, f, a.out_cand, a.cand_ld, a.out_count, a.pitch_status, a.work, unc_tol, full)) {
        if (lane == 0) a.unsure_list[atomicAdd(a.unsure_count, 1)] = (int32_t)f;
    }
}

template <int U, int W = 1>
inline size_t pow2_lds_bytes(int n, int nb, int nst = 0) {
    size_t need = (size_t)pitch_refine_lds_bytes(n, nst);
    const size_t exch = (size_t)pow2_geom<U, W>::EX * sizeof(double);
    const size_t mel = (size_t)(2 * ((nb + 1) & ~1) + 64) * sizeof(double);
    if (exch > need) need = exch;
    if (mel > need) need = mel;
    const size_t t2_end = (size_t)pow2_geom<U, W>::T2_LDS_OFFSET + pow2_geom<U, W>::T2_LDS_BYTES;
    if (t2_end > need) need = t2_end;
    return (need + 15) & ~(size_t)15;
}

void launch_refine_curve(hipStream_t s, const spectral_args_t &a, size_t lds_scan, size_t lds_refine);      // k_spectral_pow2.hip
// the split form's candidate list per frame: entries of keys / list in the refinement kernel, int32 words of the list's row in HBM
inline int spectral_split_cand_cap(int reach) { return ((reach / 4 + 8) + 7) & ~7; }
inline size_t spectral_split_list_ints(int reach) { return (size_t)(1 + (spectral_split_cand_cap(reach) + 1) / 2 + 1) & ~(size_t)1; }

// returns 1 when the call ran in the split form, 0 fused
template <int U, int W = 1>
int launch_pow2_u(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a) {
    const dim3 grid((unsigned)L.F), block(64 * W);
    a.f0 = 0; a.n_batch = L.F; a.curve = nullptr; a.curve_ld = 0; a.curve_tol = nullptr; a.curve_list = nullptr; a.list_ld = 0; a.reach = 0; a.cand_cap = 0; a.far_list = nullptr;
    a.pp.ncurve = (!L.whole_curve && L.out_r == nullptr && !L.mfcc_only) ? pitch_curve_entries(L.n, L.sample_rate, L.fmin) : 0;
    const size_t base = pow2_lds_bytes<U, W>(L.n, L.nb, a.pp.ncurve), extra = pitch_full_list_bytes(L.n, L.kmax);
    a.pp.full_off = extra ? (int)base : 0;
    const size_t lds = base + extra;
    const size_t lds_mfcc = pow2_lds_bytes<U, W>(0, L.nb), lds_ac = pow2_lds_bytes<U, W>(0, 0);
    constexpr int NC = pow2_geom<U, W>::NC;
    const bool lpc = L.out_lpc != nullptr, mf = L.out_mfcc != nullptr;
    if (L.mfcc_only) {                                       // n == Nc, or (L.n == 2 Nc) the unpadded form
        if (L.n == 2 * NC) {
            if constexpr (U * W <= 2) hipLaunchKernelGGL((analyze_pow2_kernel<U, false, true, true, SP_MFCC_HALF, W>), grid, block, lds_mfcc, s, a);
        } else if (L.interp && L.n != NC) {                  // a padded frame with interpolated bins
            a.ip = L.ip;
            const size_t li = lds_mfcc > (size_t)L.ip.lds_bytes ? lds_mfcc : (((size_t)L.ip.lds_bytes + 15) & ~(size_t)15);
            hipLaunchKernelGGL((analyze_pow2_kernel<U, false, true, false, SP_MFCC_ONLY_INTERP, W>), grid, block, li, s, a);
        } else hipLaunchKernelGGL((analyze_pow2_kernel<U, false, true, true, SP_MFCC_ONLY, W>), grid, block, lds_mfcc, s, a);
        return 0;
    }
    if (L.out_r != nullptr) {                                // autocorrelate(n_lags) alone
        if (L.n == NC) hipLaunchKernelGGL((analyze_pow2_kernel<U, false, false, true, SP_AC_ONLY, W>), grid, block, lds_ac, s, a);
        else hipLaunchKernelGGL((analyze_pow2_kernel<U, false, false, false, SP_AC_ONLY, W>), grid, block, lds_ac, s, a);
        return 0;
    }
    // The 4096-point plan in separate kernels (vbx_spectral.hpp, SP_ANALYZE_SPLIT; k_spectral_pow2.hip): transforms + LPC + MFCC per batch of
    // frames, the lag curves through a scratch buffer, then the peak scan, the refinement at eleven frames per CU, the far frames.  Where the curve is cut (an even frame length at speech settings),
    // the caller gave scratch and kmax needs no list region in LDS.
    if constexpr (U * W == 4) {
        // an odd frame length keeps its whole curve (a.pp.ncurve == 0: pitch_curve_entries) -- in the scratch row too, for the whole-curve
        // kernel of the far frames; the scan and the refinement kernel need no more of it than of an even length's
        const bool odd_ok = (L.n & 1) && !L.whole_curve && L.out_r == nullptr && !L.mfcc_only;
        const int nst_row = a.pp.ncurve > 0 ? a.pp.ncurve : (odd_ok ? L.n + 1 : 0);
        const size_t row_doubles = (size_t)(nst_row + Y_PAD);
        const bool full = L.n == NC;
        // Measured (tools/experiments/split_check.py, 2 h of audio per step, pipeline M frames/s fused -> split, all bit-identical):
        // 2050 / 1024: 11.7 -> 13.3, 2500 / 1000: 11.1 -> 12.5, 3000 / 1200: 10.8 -> 11.9, 4000 / 2000: 9.3 -> 10.2, 4096 / 2048: 11.2 -> 12.1,
        // 4096 / 1024: 11.3 -> 12.5; pitch at kmax = 8 (40,000 frames): 13.9 -> 9.1 ms.  Per frame at 4096 / 2048: transforms 37.3 ns,
        // scan 4.3, refinement 21.0 (eleven frames per CU), far frames 2.1 -- against 69 ns fused.  (A first form without the scan kernel
        // -- the whole curve and the full-size candidate list in the refinement kernel's LDS, six frames per CU at 4096 samples -- lost
        // to the fused kernel from 3,800 samples on.)
        const int reach = pitch_curve_reach(L.n, L.sample_rate, L.fmin);
        const bool want = L.curve_ws != nullptr && nst_row > 0 && reach > 0 && extra == 0 && (full || !mf || L.interp);
        const size_t list_ints = spectral_split_list_ints(reach > 0 ? reach : 16);
        const size_t cap = (want && L.curve_ws_bytes > 64) ? (L.curve_ws_bytes - 64) / ((row_doubles + 1) * sizeof(double) + (list_ints + 1) * sizeof(int32_t)) : 0;
        if (want && cap >= 1024) {
            a.ip = L.ip;
            size_t la = pow2_lds_bytes<U, W>(0, mf ? L.nb : 0);
            if (mf && !full && (size_t)L.ip.lds_bytes > la) la = ((size_t)L.ip.lds_bytes + 15) & ~(size_t)15;
            a.reach = reach; a.cand_cap = spectral_split_cand_cap(reach);
            const size_t ls = ((size_t)pitch_refine_lds_bytes(L.n, a.pp.ncurve) + 16 + 15) & ~(size_t)15;      // (the whole-curve kernel: + the odd row's pair partner)
            const size_t lr = ((size_t)pitch_refine_lds_bytes(L.n, reach, a.cand_cap) + 15) & ~(size_t)15;
            a.curve = L.curve_ws; a.curve_ld = (long)row_doubles; a.curve_tol = L.curve_ws + cap * row_doubles;
            a.curve_list = reinterpret_cast<int32_t *>(a.curve_tol + cap); a.list_ld = (long)list_ints;
            a.far_list = a.curve_list + cap * list_ints;     // [4 + cap] int32
            for (long f0 = 0; f0 < L.F; f0 += (long)cap) {
                a.f0 = f0; a.n_batch = (L.F - f0 < (long)cap) ? L.F - f0 : (long)cap;
                const dim3 g((unsigned)a.n_batch);
                if (full) {
                    if (mf && lpc) hipLaunchKernelGGL((analyze_pow2_kernel<U, true, true, true, SP_ANALYZE_SPLIT, W>), g, block, la, s, a);
                    else if (mf) hipLaunchKernelGGL((analyze_pow2_kernel<U, false, true, true, SP_ANALYZE_SPLIT, W>), g, block, la, s, a);
                    else if (lpc) hipLaunchKernelGGL((analyze_pow2_kernel<U, true, false, true, SP_ANALYZE_SPLIT, W>), g, block, la, s, a);
                    else hipLaunchKernelGGL((analyze_pow2_kernel<U, false, false, true, SP_ANALYZE_SPLIT, W>), g, block, la, s, a);
                } else if (mf && lpc) hipLaunchKernelGGL((analyze_pow2_kernel<U, true, true, false, SP_ANALYZE_INTERP_SPLIT, W>), g, block, la, s, a);
                else if (mf) hipLaunchKernelGGL((analyze_pow2_kernel<U, false, true, false, SP_ANALYZE_INTERP_SPLIT, W>), g, block, la, s, a);
                else if (lpc) hipLaunchKernelGGL((analyze_pow2_kernel<U, true, false, false, SP_ANALYZE_SPLIT, W>), g, block, la, s, a);
                else hipLaunchKernelGGL((analyze_pow2_kernel<U, false, false, false, SP_ANALYZE_SPLIT, W>), g, block, la, s, a);
                launch_refine_curve(s, a, ls, lr);
            }
            return 1;
        }
    }
    if (L.interp && mf && L.n != NC) {         // a padded frame whose MFCC bins are interpolated from the transform's
        a.ip = L.ip;
        const size_t li = lds > (size_t)L.ip.lds_bytes ? lds : (((size_t)L.ip.lds_bytes + 15) & ~(size_t)15);
        if (lpc) hipLaunchKernelGGL((analyze_pow2_kernel<U, true, true, false, SP_ANALYZE_INTERP, W>), grid, block, li, s, a);
        else hipLaunchKernelGGL((analyze_pow2_kernel<U, false, true, false, SP_ANALYZE_INTERP, W>), grid, block, li, s, a);
        return 0;
    }
    if (L.n != NC) {                           // a padded frame; MFCC from the transform's own bins when its length divides M (U = 1: 512)
        if constexpr (U * W == 1) {
            if (lpc && mf) { hipLaunchKernelGGL((analyze_pow2_kernel<U, true, true, false, SP_ANALYZE, W>), grid, block, lds, s, a); return 0; }
            if (mf) { hipLaunchKernelGGL((analyze_pow2_kernel<U, false, true, false, SP_ANALYZE, W>), grid, block, lds, s, a); return 0; }
        }
        if (lpc) hipLaunchKernelGGL((analyze_pow2_kernel<U, true, false, false, SP_ANALYZE, W>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((analyze_pow2_kernel<U, false, false, false, SP_ANALYZE, W>), grid, block, lds, s, a);
    } else if (lpc && mf) hipLaunchKernelGGL((analyze_pow2_kernel<U, true, true, true, SP_ANALYZE, W>), grid, block, lds, s, a);
    else if (lpc) hipLaunchKernelGGL((analyze_pow2_kernel<U, true, false, true, SP_ANALYZE, W>), grid, block, lds, s, a);
    else if (mf) hipLaunchKernelGGL((analyze_pow2_kernel<U, false, true, true, SP_ANALYZE, W>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((analyze_pow2_kernel<U, false, false, true, SP_ANALYZE, W>), grid, block, lds, s, a);
    return 0;
}


}  // namespace vbx
