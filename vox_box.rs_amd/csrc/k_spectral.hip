// k_spectral.hip -- one spectral pass per frame feeding pitch, LPC and MFCC: the transform of 2400 real points, for frames of
// 1200 samples (25 ms at 48 kHz) and, zero padded, of 1025..1199 (k_spectral_pow2*.hip hold the 1024 / 2048 / 4096 forms).
// The same kernel also serves MFCC::mfcc alone and Autocorrelate::autocorrelate with many lags (MODE), and this file holds
// the host side shared by all forms: which transform serves a frame length (spectral_plan), the twiddle tables, the launch.
//
// Reference rows served: A1-A3 (autocorrelate, normalize, lag window), A4-A9 (pitch, through
// vbx_pitch_refine.hpp), A10 (lpc on r[0..12]), A14 (mfcc).  What the reference computes with an O(N^2) fold per lag
// (src/periodic.rs:276-289) and a separate rustfft call (src/spectrum.rs:416-419) is here ONE real FFT of the
// zero-padded windowed frame, length M = 2N = 2400:
//     X = FFT_M(x_w padded)                      |X[k]|^2 -> inverse FFT -> S[lag] = sum_i x[i] x[i+lag], every lag
//     X[2k'] = the N-point DFT bin k' of the frame  -> the |X|^2 and |X| the mel filters of MFCC::mfcc read
//     S[0..12]                                      -> the lag sums of LPC::lpc (round 6: the recursion itself, like MFCC's log10 + DCT,
//                                                      runs afterwards in levinson_rows_kernel_t, k_lpc.hip: a lane per record)
// (Q1: the reference's fold is seeded with x[0], r[lag] = S[lag] - x0*x[lag] + x0; applied afterwards.)  This replaces
// 1.44 MFLOP of autocorrelation MACs per frame by about 0.3 MFLOP and removes the MFCC and LPC kernels' passes over the
// same frame.  Accuracy: forward + inverse f64 FFT, error ~1e-16 * S[0] per lag (measured against the oracle in
// tests/test_gpu_parity.py), far inside the 1e-6 relative tolerance of the autocorrelation / LPC / MFCC rows.
//
// One wavefront per frame, the transform lives in registers:
//   real FFT by the packing trick: z[j] = x[2j] + i x[2j+1], complex FFT of length N_c = 1200 = 20 * 20 * 3, then the
//   split into the spectrum of the real sequence; the inverse likewise with the roles exchanged.
//   complex FFT (decimation in frequency, n = 60 a + 3 b + c, k = ka + 20 kb + 400 kc):
//     stage 1  lane n' = 3b + c (60 lanes): 20-point DFT over a in registers (4 x 5 prime-factor form, no twiddles
//              inside), times W_1200^(n' ka)
//     stage 2  lane (ka, c) (60 lanes): 20-point DFT over b, times W_60^(c kb)
//     stage 3  lane q = ka + 20 kb, 7 per lane: 3-point DFT over c -> X[q + 400 kc], natural order
//   between the stages the values change lanes through LDS, real and imaginary parts in two passes (one 1200-double
//   buffer, which the lag curve y later overwrites: the frame state stays at 13.5 KB = 12 wavefronts per CU).
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"
#include "vbx_mfcc_tail.hpp"
#include "vbx_mfcc_interp.hpp"
#include "vbx_pitch_refine.hpp"
#include "vbx_spectral.hpp"

namespace vbx {

constexpr int SP_N = SPECTRAL_N;             // frame length = complex FFT length
constexpr int SP_M = 2 * SP_N;               // real FFT length
constexpr int SP_S1 = 61;                    // exchange 1 row stride [ka][n']   (odd: the 20 rows hit distinct banks)
constexpr int SP_S2 = 404;                   // exchange 2 row stride [c][ka + 20 kb]
constexpr int SP_T1 = 0;                     // twiddle table (complex entries): T1[60][20] = W_1200^(n' ka)
constexpr int SP_T2 = SP_T1 + 60 * 20;       //                                  T2[3][20]  = W_60^(c kb)
constexpr int SP_TM = SP_T2 + 3 * 20;        //                                  WM[601]    = W_2400^m
static_assert(SP_TM + 601 == SPECTRAL_TAB_COMPLEX, "table layout");
// LDS: [0, 9760) exchange buffer (later the lag curve), [0, 10144) the mel sums between the transforms, then the copy of T2
constexpr int SP_T2_LDS_OFFSET = (2 * 602 + 64) * 8;
static_assert(SP_T2_LDS_OFFSET >= (20 * SP_S1 > 3 * SP_S2 ? 20 * SP_S1 : 3 * SP_S2) * 8 && SP_T2_LDS_OFFSET % 16 == 0, "T2 behind both");

__device__ __forceinline__ void dft5(double &r0, double &i0, double &r1, double &i1, double &r2, double &i2,
                                     double &r3, double &i3, double &r4, double &i4) {
    constexpr double C5 = 0.55901699437494742410;    // (cos 72 - cos 144) / 2
    constexpr double S1 = 0.95105651629515357212;    // sin 72
    constexpr double S2 = 0.58778525229247312917;    // sin 144
    const double t1r = r1 + r4, t1i = i1 + i4, t3r = r1 - r4, t3i = i1 - i4;
    const double t2r = r2 + r3, t2i = i2 + i3, t4r = r2 - r3, t4i = i2 - i3;
    const double t5r = t1r + t2r, t5i = t1i + t2i;
    const double m1r = fma(-0.25, t5r, r0), m1i = fma(-0.25, t5i, i0);
    const double m2r = C5 * (t1r - t2r), m2i = C5 * (t1i - t2i);
    const double s1r = m1r + m2r, s1i = m1i + m2i, s2r = m1r - m2r, s2i = m1i - m2i;
    // u = S1 t3 + S2 t4, v = S2 t3 - S1 t4;  X1 = s1 - i u, X4 = s1 + i u, X2 = s2 - i v, X3 = s2 + i v
    const double ur = fma(S1, t3r, S2 * t4r), ui = fma(S1, t3i, S2 * t4i);
    const double vr = fma(S2, t3r, -(S1 * t4r)), vi = fma(S2, t3i, -(S1 * t4i));
    r0 = r0 + t5r; i0 = i0 + t5i;
    r1 = s1r + ui; i1 = s1i - ur;
    r4 = s1r - ui; i4 = s1i + ur;
    r2 = s2r + vi; i2 = s2i - vr;
    r3 = s2r - vi; i3 = s2i + vr;
}

// 20-point DFT in place, prime-factor form (gcd(4, 5) = 1: no twiddles between the 4- and the 5-point parts).
// Input index a sits in slot a; output index k is left in slot dft20_slot(k).
__host__ __device__ constexpr int dft20_in(int n1, int n2) { return (5 * n1 + 4 * n2) % 20; }
__host__ __device__ constexpr int dft20_slot(int k) { return (5 * (k % 4) + 4 * (k % 5)) % 20; }

__device__ __forceinline__ void dft20(double (&re)[20], double (&im)[20]) {
#pragma unroll
    for (int n2 = 0; n2 < 5; n2++)
        dft4(re[dft20_in(0, n2)], im[dft20_in(0, n2)], re[dft20_in(1, n2)], im[dft20_in(1, n2)],
             re[dft20_in(2, n2)], im[dft20_in(2, n2)], re[dft20_in(3, n2)], im[dft20_in(3, n2)]);
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
        dft5(re[dft20_in(k1, 0)], im[dft20_in(k1, 0)], re[dft20_in(k1, 1)], im[dft20_in(k1, 1)],
             re[dft20_in(k1, 2)], im[dft20_in(k1, 2)], re[dft20_in(k1, 3)], im[dft20_in(k1, 3)],
             re[dft20_in(k1, 4)], im[dft20_in(k1, 4)]);
}

// Complex FFT of length 1200.  In: lane n' < 60 holds z[60 a + n'] in (re[a], im[a]).  Out: lane l holds
// X[l + 64 t + 400 kc] in (xr[t][kc], xi[t][kc]) for l + 64 t < 400 (KC = 3: every output; KC = 2: kc = 0, 1 only).
// ex: LDS exchange buffer (>= 20 * SP_S1 doubles).  tab: twiddle table.
// The twenty twiddle products of a stage in batches of TWB (two), each batch finished before the next one's loads may start: the
// products are pinned (an empty asm with the value as in/out operand) and the loads fenced (a compiler memory barrier).  Left
// alone the compiler requests all twenty twiddles at once -- 80 registers -- right after the 20-point DFT, whose results it
// spills to make room (the 168-register instance: ~100 scratch round trips per transform).  Batches of 2 / 5 / 10: 0 / 12 / 9
// spilled registers in the fused kernel, 34.1 / 34.3 / 34.6 M frames/s -- two it is: no scratch traffic at all.
#ifndef VBX_EXP_TWB
#define VBX_EXP_TWB 2
#endif
constexpr int TWB = VBX_EXP_TWB;
#ifndef VBX_EXP_MB1
#define VBX_EXP_MB1 3
#endif
template <int TWB = vbx::TWB>
__device__ __forceinline__ void twiddle_tight(double (&re)[20], double (&im)[20], const double2 *tw_row) {
#pragma unroll
    for (int h = 0; h < 20 / TWB; h++) {
        double2 tw[TWB];
#pragma unroll
        for (int k = 0; k < TWB; k++) tw[k] = tw_row[TWB * h + k];
#pragma unroll
        for (int k = 0; k < TWB; k++) {
            if (TWB * h + k == 0) continue;
            const int s = dft20_slot(TWB * h + k);
            const double a = re[s], b = im[s];
            re[s] = fma(a, tw[k].x, -(b * tw[k].y));
            im[s] = fma(a, tw[k].y, b * tw[k].x);
            asm volatile("" : "+v"(re[s]), "+v"(im[s]));
        }
        asm volatile("" ::: "memory");
    }
}

// TIGHT (the instance compiled for three wavefronts per SIMD, 168 registers): the scheduler may not move the twiddle loads
// above the 20-point DFT they follow -- hoisted there to hide their latency they hold 40..80 registers while the DFT needs
// them, and the DFT's own values spill (scratch round trips inside both transforms).
// t2: the stage-2 twiddles T2[3][20] (960 B) in LDS (analyze_kernel copies them there once per frame, into a region that only
// the refinement uses later): ten dependent round trips to the L1 / L2 per transform become LDS reads.  (Round 5: s_memtime
// at the phase boundaries showed a wavefront spending 23 % of its life in the two transforms and the split between them
// for 16 % of its instructions -- twenty batches of two twiddle loads per transform, each waited for.)
template <int KC, bool TIGHT = false>
__device__ __forceinline__ void fft1200(double (&re)[20], double (&im)[20], double (&xr)[7][3], double (&xi)[7][3],
                                        double *ex, const double2 *tab, const double2 *t2) {
    const int lane = lane_id();
    const int np = (lane < 60) ? lane : 59;                 // lanes 60..63 shadow lane 59 (they never write)
    const bool act = lane < 60;
    // stage 1 (the twiddles arrive in two batches of ten: the 20-point DFT needs the registers)
    dft20(re, im);
    if constexpr (TIGHT) twiddle_tight(re, im, tab + SP_T1 + np * 20);
    else {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        double2 tw[10];
#pragma unroll
        for (int k = 0; k < 10; k++) tw[k] = tab[SP_T1 + np * 20 + 10 * h + k];
#pragma unroll
        for (int k = 0; k < 10; k++) {
            if (10 * h + k == 0) continue;
            const int s = dft20_slot(10 * h + k);
            const double a = re[s], b = im[s];
            re[s] = fma(a, tw[k].x, -(b * tw[k].y));
            im[s] = fma(a, tw[k].y, b * tw[k].x);
        }
    }
    }
    // exchange 1: [ka][n'] -> lane (ka2, c2) = (lane % 20, lane / 20) reads n' = 3 b + c2
    const int ka2 = np % 20, c2 = np / 20;
    double br[20], bi[20];
    wave_sync();
#pragma unroll
    for (int k = 0; k < 20; k++) if (act) ex[k * SP_S1 + lane] = re[dft20_slot(k)];
    wave_sync();
#pragma unroll
    for (int b = 0; b < 20; b++) br[b] = ex[ka2 * SP_S1 + 3 * b + c2];
    wave_sync();
#pragma unroll
    for (int k = 0; k < 20; k++) if (act) ex[k * SP_S1 + lane] = im[dft20_slot(k)];
    wave_sync();
#pragma unroll
    for (int b = 0; b < 20; b++) bi[b] = ex[ka2 * SP_S1 + 3 * b + c2];
    // stage 2
    dft20(br, bi);
    if constexpr (TIGHT) twiddle_tight<4>(br, bi, t2 + c2 * 20);
    else {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        double2 tw[10];
#pragma unroll
        for (int k = 0; k < 10; k++) tw[k] = t2[c2 * 20 + 10 * h + k];
#pragma unroll
        for (int k = 0; k < 10; k++) {
            if (10 * h + k == 0) continue;
            const int s = dft20_slot(10 * h + k);
            const double a = br[s], b = bi[s];
            br[s] = fma(a, tw[k].x, -(b * tw[k].y));
            bi[s] = fma(a, tw[k].y, b * tw[k].x);
        }
    }
    }
    // exchange 2: [c][ka + 20 kb] -> lane l reads q = l + 64 t for c = 0..2
    double vr[7][3], vi[7][3];
    wave_sync();
#pragma unroll
    for (int k = 0; k < 20; k++) if (act) ex[c2 * SP_S2 + ka2 + 20 * k] = br[dft20_slot(k)];
    wave_sync();
#pragma unroll
    for (int t = 0; t < 7; t++)
#pragma unroll
        for (int c = 0; c < 3; c++) vr[t][c] = (t < 6 || lane < 16) ? ex[c * SP_S2 + lane + 64 * t] : 0.0;
    wave_sync();
#pragma unroll
    for (int k = 0; k < 20; k++) if (act) ex[c2 * SP_S2 + ka2 + 20 * k] = bi[dft20_slot(k)];
    wave_sync();
#pragma unroll
    for (int t = 0; t < 7; t++)
#pragma unroll
        for (int c = 0; c < 3; c++) vi[t][c] = (t < 6 || lane < 16) ? ex[c * SP_S2 + lane + 64 * t] : 0.0;
    wave_sync();
    // stage 3: X[kc] = v0 + v1 W3^kc + v2 W3^(2 kc)
    constexpr double H3 = 0.86602540378443864676;           // sin 60
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const double sr = vr[t][1] + vr[t][2], si = vi[t][1] + vi[t][2];
        const double dr = vr[t][1] - vr[t][2], di = vi[t][1] - vi[t][2];
        xr[t][0] = vr[t][0] + sr; xi[t][0] = vi[t][0] + si;
        const double mr = fma(-0.5, sr, vr[t][0]), mi = fma(-0.5, si, vi[t][0]);
        xr[t][1] = fma(H3, di, mr); xi[t][1] = fma(-H3, dr, mi);           // m - i H3 d
        if (KC == 3) { xr[t][2] = fma(-H3, di, mr); xi[t][2] = fma(H3, dr, mi); }
        else { xr[t][2] = 0.0; xi[t][2] = 0.0; }
    }
}

// WAVES: wavefronts per SIMD the instance is compiled for.  Two: the transforms take ~230 registers.  Three (168 registers, what
// launch_analyze picks since the end of round 4): with the twiddle products in pinned batches (twiddle_tight) nothing spills in
// the fused kernel; before that ~55 registers did and the third wavefront cost more than it brought.
#ifndef VBX_SPECTRAL_WAVES
#define VBX_SPECTRAL_WAVES 2
#endif
// FULL: the frame fills the transform (n == 1200, the bounds tests fold away); otherwise n < 1200, zero padded: 1025..1199,
// and 600 / 800 when MFCC is wanted (they divide M = 2400: spectral_plan_mfcc).
// MODE: SP_ANALYZE the fused analysis; SP_MFCC_ONLY MFCC::mfcc alone (vbx_mfcc_f64 on a full frame): the forward transform and
// the mel / DCT tail, nothing after them; SP_AC_ONLY Autocorrelate::autocorrelate alone (vbx_autocorrelate_f64 with many lags):
// both transforms, the fold seed, the lag sums stored.
// WAVES: wavefronts per SIMD the kernel is compiled for.  2 (the default: no spills) for kmax = 1, where the two transforms
// are a third of the kernel; 3 for 2 <= kmax <= 64, where the kernel is almost all refinement -- chains of dependent FP64
// operations that two wavefronts do not cover (measured at kmax = 2 / 8 / 64: 12.1 -> 13.6, 4.55 -> 5.65, 2.2 -> 2.7 M
// frames/s; kmax = 1: 32.5 -> 31.7, hence the split).  The transforms then spill ~55 registers, which the refinement hides.
template <bool LPC, bool MFCC, bool FULL, int MODE = SP_ANALYZE, int WAVES = VBX_SPECTRAL_WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sp_is_mfcc_only(MODE) ? 2 : WAVES, sp_is_mfcc_only(MODE) ? 4 : WAVES))) void analyze_kernel(const spectral_args_t a) {
    static_assert(MODE != SP_MFCC_ONLY || (MFCC && FULL && !LPC), "the MFCC-only form needs the full frame and has no lag sums");
    static_assert(MODE != SP_AC_ONLY || (!MFCC && !LPC), "the autocorrelation-only form");
    static_assert(!sp_is_interp(MODE) || (MFCC && !FULL), "interpolated bins: a padded frame's MFCC");
    static_assert(MODE != SP_MFCC_ONLY_INTERP || !LPC, "the MFCC-only forms have no lag sums");
    constexpr bool PITCH = !sp_is_mfcc_only(MODE);           // the second transform runs
    constexpr bool INTERP = sp_is_interp(MODE);              // MFCC's bins lie between the transform's (mfcc_interp_t, vbx_kernels.hpp)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long f = xcd_item(blockIdx.x, a.n_frames);            // neighbouring frames on the same XCD: their overlap hits its L2
    if (f >= a.n_frames) return;
    const int lane = lane_id();
    const int np = (lane < 60) ? lane : 59;
    const int n = FULL ? SP_N : a.n;                         // frame length, <= SP_N (shorter: longer zero padding)
    double *ex = smem;                                       // exchange buffer, later the lag curve y
    const double *xf = a.frames + f * a.stride;
    // stage-2 twiddles into LDS, behind the exchange buffer and the mel sums (fft1200; ordered by the first exchange's wave_sync)
    double2 *t2 = reinterpret_cast<double2 *>(reinterpret_cast<char *>(smem) + SP_T2_LDS_OFFSET);
    VBX_PHASE_INIT();

    // ---- load: z[60 a + n'] = (xw[120 a + 2 n'], xw[120 a + 2 n' + 1]), a < 10 (the rest is the zero padding) ----
    double re[20], im[20];
    const int16_t *x16 = reinterpret_cast<const int16_t *>(a.frames) + f * a.stride;       // the frame when a.pcm (FULL only)
    {
        const bool al = ((((uintptr_t)xf) | ((uintptr_t)a.window)) & 15) == 0;      // uniform
        double2 xv[10], wv[10];
#pragma unroll
        for (int q = 0; q < 10; q++) {
            const int i = 120 * q + 2 * np;
            if (FULL && (a.pcm & SP_FLAG_PCM)) {
                // 16-bit PCM in: 960 B of new samples per frame instead of 3840 (the host-fed case: PCIe carries the PCM);
                // widened here exactly as vbx_pcm16_to_f64 would have (bit-identical frames, tests/test_gpu_frontend.py)
                int lo, hi;
                if ((((uintptr_t)x16) & 3) == 0) { const int w = *reinterpret_cast<const int *>(x16 + i); lo = (short)(w & 0xffff); hi = w >> 16; }
                else { lo = x16[i]; hi = x16[i + 1]; }
                xv[q] = double2{pcm16_value(lo), pcm16_value(hi)};
                wv[q] = (a.window != nullptr) ? (((uintptr_t)a.window & 15) == 0 ? *reinterpret_cast<const double2 *>(a.window + i)
                                                                                  : double2{a.window[i], a.window[i + 1]}) : double2{1.0, 1.0};
            } else if (al && i + 1 < n) {
                xv[q] = *reinterpret_cast<const double2 *>(xf + i);
                wv[q] = (a.window != nullptr) ? *reinterpret_cast<const double2 *>(a.window + i) : double2{1.0, 1.0};
            } else {
                xv[q] = double2{0.0, 0.0}; wv[q] = double2{1.0, 1.0};
                if (i < n) { xv[q].x = xf[i]; if (a.window != nullptr) wv[q].x = a.window[i]; }
                if (i + 1 < n) { xv[q].y = xf[i + 1]; if (a.window != nullptr) wv[q].y = a.window[i + 1]; }
            }
        }
#pragma unroll
        for (int q = 0; q < 10; q++) {
            re[q] = (a.window != nullptr) ? xv[q].x * wv[q].x : xv[q].x;
            im[q] = (a.window != nullptr) ? xv[q].y * wv[q].y : xv[q].y;
        }
#pragma unroll
        for (int q = 10; q < 20; q++) { re[q] = 0.0; im[q] = 0.0; }
    }
    const double x0 = readlane_f64(re[0], 0);               // x_w[0], for the fold seed (Q1)
    if (lane < 60) t2[lane] = a.tab[SP_T2 + lane];          // (here, not before the frame's loads: the registers are fewest)
    VBX_PHASE(a.work, f, 0);

    // ---- forward transform of the packed frame ----
    double xr[7][3], xi[7][3];
    fft1200<3, (WAVES >= 3)>(re, im, xr, xi, ex, a.tab, t2);
    VBX_PHASE(a.work, f, 1);

    // ---- exchange 3: natural order, then each lane takes the pairs (m, N - m), m = lane + 64 t <= 600 ----
    double ar[10], ai[10], br[10], bi[10];
#pragma unroll
    for (int t = 0; t < 7; t++)
#pragma unroll
        for (int kc = 0; kc < 3; kc++) if (t < 6 || lane < 16) ex[lane + 64 * t + 400 * kc] = xr[t][kc];
    wave_sync();
#pragma unroll
    for (int t = 0; t < 10; t++) {
        const int m = lane + 64 * t;
        const bool ok = m <= 600;
        ar[t] = ok ? ex[m] : 0.0;
        br[t] = ok ? ex[(m == 0) ? 0 : SP_N - m] : 0.0;
    }
    wave_sync();
#pragma unroll
    for (int t = 0; t < 7; t++)
#pragma unroll
        for (int kc = 0; kc < 3; kc++) if (t < 6 || lane < 16) ex[lane + 64 * t + 400 * kc] = xi[t][kc];
    wave_sync();
#pragma unroll
    for (int t = 0; t < 10; t++) {
        const int m = lane + 64 * t;
        const bool ok = m <= 600;
        ai[t] = ok ? ex[m] : 0.0;
        bi[t] = ok ? ex[(m == 0) ? 0 : SP_N - m] : 0.0;
    }
    wave_sync();

    // ---- spectrum of the real sequence: X[m] = E + T, X[N - m] = conj(E - T); powers; the inverse transform's input ----
    //   E = (A + conj B) / 2, O = -i (A - conj B) / 2, T = W_M^m O;   P[m] = |E + T|^2, P[N - m] = |E - T|^2
    //   the inverse's input G[m] = S - i D w, G[N - m] = S - i D conj(w)   (S = P[m] + P[N-m], D = P[m] - P[N-m], w = W_M^m)
    const int b_lo = MFCC ? a.bins[0] : 0;
    double pk[10], pn[10];                                   // P[m], P[N - m]
    double2 *zc = reinterpret_cast<double2 *>(ex);           // INTERP: Z[j - jmin] = X_M[j] e^{2 pi i j c / M}, Z[-j] = conj Z[j]
    double2 rot_m = double2{1.0, 0.0}, rot_step = double2{1.0, 0.0};
    if constexpr (INTERP) { rot_m = reinterpret_cast<const double2 *>(a.ip.rot)[lane]; rot_step = reinterpret_cast<const double2 *>(a.ip.rot)[64]; }
#pragma unroll
    for (int t = 0; t < 10; t++) {
        const int m = lane + 64 * t;
        const double2 w = a.tab[SP_TM + ((m <= 600) ? m : 0)];
        const double er = 0.5 * (ar[t] + br[t]), ei = 0.5 * (ai[t] - bi[t]);
        const double o_r = 0.5 * (ai[t] + bi[t]), o_i = -0.5 * (ar[t] - br[t]);
        const double tr = fma(w.x, o_r, -(w.y * o_i)), ti = fma(w.x, o_i, w.y * o_r);
        const double pr = er + tr, pi = ei + ti, qr = er - tr, qi = ei - ti;
        pk[t] = fma(pr, pr, pi * pi);
        pn[t] = fma(qr, qr, qi * qi);
        if constexpr (INTERP) {                              // (every lane is past exchange 3's last read: the buffer is free)
            asm volatile("" : "+v"(pk[t]), "+v"(pn[t]));     // the powers NOW: two values wait for exchange 4, not the four they are made of
            mfcc_interp_stage(zc, a.ip, m, pr, pi, rot_m, rot_step, t == 0);
        }
    }

    // ---- MFCC::mfcc at a length that does not divide the transform: each of the frame's DFT bins from 24 .. 40 of the
    //      transform's (lane l: bins b_lo + l + 64 u), BEFORE exchange 4 takes the buffer; then the same products and tail ----
    if constexpr (INTERP) {
        wave_sync();
        VBX_PHASE(a.work, f, 13);
        const int nbp = (a.nb + 1) & ~1;
        double *pu = ex + a.ip.pu_off, *pd = pu + nbp, *en = pd + nbp;
        const double2 *cf = reinterpret_cast<const double2 *>(a.ip.coef) + lane;
        const int HT = a.ip.taps >> 1;                       // 12, 16 or 20 pairs of taps (the host's choice for M / n)
        for (int u = 0; u * 64 < a.nb; u++) {
            const int b = lane + 64 * u;
            const double2 *zp = zc + a.ip.j0[u * 64 + lane];
            const double2 sl = *reinterpret_cast<const double2 *>(a.slopes + 2 * ((b < a.nb) ? b : 0));
            double vr, vi;
            mfcc_interp_bin(HT, cf + (u * HT) * 64, 64, zp, vr, vi);
            const double pw = fma(vr, vr, vi * vi);
            if (b < a.nb) {
                pu[b] = fabs(pw) * sl.x;                     // norm_sqr * multiplier (src/spectrum.rs:426-428)
                pd[b] = fabs(sqrt(pw)) * sl.y;               // norm * multiplier (:432-434)
            }
        }
        wave_sync();
        VBX_PHASE(a.work, f, 14);
        double2 t2v = double2{0.0, 0.0};                     // the products may lie over the stage-2 twiddles: requested now, put back after the tail
        if constexpr (PITCH) t2v = a.tab[SP_T2 + np];
        if (a.num_coeffs <= 16) mfcc_tail_q(pu, pd, en, a.bins, a.dct, a.num_coeffs, b_lo, lane, a.out_mfcc + f * a.mfcc_ld, a.work, f, (a.pcm & SP_FLAG_MFCC_DEFER) != 0);
        else mfcc_tail_m(pu, pd, en, a.bins, a.dct, a.num_coeffs, b_lo, lane, a.out_mfcc + f * a.mfcc_ld);
        if (a.mfcc_status != nullptr && lane == 0) a.mfcc_status[f] = 0;
        wave_sync();
        if (PITCH && lane < 60) t2[lane] = t2v;
        VBX_PHASE(a.work, f, 15);
    }

    if constexpr (PITCH) {
        // ---- exchange 4: G in natural order -> stage-1 layout of the second transform ----
        // (the ten twiddles W_M^m requested together and without a condition -- index 0 stands in past m = 600 --, not one
        // by one behind `if (m <= 600)`: a load inside a branch cannot be moved out of it, and each waited for its own)
        double2 wm[10];
    #pragma unroll
        for (int t = 0; t < 10; t++) { const int m = lane + 64 * t; wm[t] = a.tab[SP_TM + ((m <= 600) ? m : 0)]; }
    #pragma unroll
        for (int t = 0; t < 10; t++) {
            const int m = lane + 64 * t;
            if (m <= 600) {
                const double2 w = wm[t];
                const double sm = pk[t] + pn[t], d = pk[t] - pn[t];
                ex[m] = fma(d, w.y, sm);
                if (m >= 1 && m < 600) ex[SP_N - m] = fma(-d, w.y, sm);
            }
        }
        wave_sync();
    #pragma unroll
        for (int q = 0; q < 20; q++) re[q] = ex[60 * q + np];
        wave_sync();
    #pragma unroll
        for (int t = 0; t < 10; t++) {
            const int m = lane + 64 * t;
            if (m <= 600) {
                const double2 w = wm[t];
                const double gi = -((pk[t] - pn[t]) * w.x);
                ex[m] = gi;
                if (m >= 1 && m < 600) ex[SP_N - m] = gi;
            }
        }
        wave_sync();
    #pragma unroll
        for (int q = 0; q < 20; q++) im[q] = ex[60 * q + np];
        wave_sync();
    }

    VBX_PHASE(a.work, f, 2);
    // ---- MFCC::mfcc from the powers: the frame's n-point DFT bin k' is X_M[q k'], q = M / n (2 for the full frame; a
    //      shorter frame whose length divides M = 2400 -- 800, 600 -- is zero padded and its bins are every q-th one): bin m / q
    //      from P[m] and bin n/2 - m / q from P[N - m] ----
    if constexpr (MFCC && !INTERP) {
        const int nbp = (a.nb + 1) & ~1;
        const int q = FULL ? 2 : a.mfcc_q, half = FULL ? SP_N / 2 : a.n / 2;
        double *pu = ex, *pd = ex + nbp, *en = ex + 2 * nbp; // the exchange buffer is free between the two transforms
        constexpr int MB = 2;     // slots per batch (five: 24 registers spilled in the three-wavefront instance)
        // Can a mirrored bin n/2 - m/q (from P[N - m]) be one of the filters' at all?  Only when they reach above a quarter of
        // the sampling rate (m <= 600: n/2 - m/q >= 600/q).  Below that -- speech settings: 8 kHz of 24 -- only P[m] has bins,
        // half as many slope pairs are wanted, and five slots' pairs are requested together instead of two: two round trips
        // to the L2 per frame instead of five (the phase clocks: 8 k cycles of the frame's 170 k for ~100 instructions).
        const bool two_sided = (half - 600 / q) - b_lo < a.nb;
        if (!two_sided) {
            constexpr int MB1 = VBX_EXP_MB1;               // (batches of three: nothing spills, 18.37 ms per 720,000 frames; of five: six registers, 18.45; round 4's form: 18.56)
#pragma unroll
            for (int h = 0; h < (10 + MB1 - 1) / MB1; h++) {
                double2 s1[MB1];
                int c1[MB1];
#pragma unroll
                for (int u = 0; u < MB1; u++) {
                    const int t = MB1 * h + u, m = lane + 64 * t;
                    if (t >= 10) { c1[u] = -1; continue; }
                    const bool on = m <= 600 && (FULL ? (m & 1) == 0 : m % q == 0);
                    const int b1 = (FULL ? (m >> 1) : m / q) - b_lo;
                    c1[u] = (on && b1 >= 0 && b1 < a.nb) ? b1 : -1;
                    s1[u] = *reinterpret_cast<const double2 *>(a.slopes + 2 * (c1[u] < 0 ? 0 : c1[u]));
                }
#pragma unroll
                for (int u = 0; u < MB1; u++) {
                    const int t = MB1 * h + u;
                    if (t < 10 && c1[u] >= 0) {
                        pu[c1[u]] = fabs(pk[t]) * s1[u].x;   // norm_sqr * multiplier (src/spectrum.rs:426-428)
                        pd[c1[u]] = fabs(sqrt(pk[t])) * s1[u].y;   // norm * multiplier (:432-434)
                    }
                }
            }
        } else
        // (the slope pairs of a few slots requested together, without a condition -- pair 0 stands in for a slot
        // without a bin --, then the products: behind `if (bin in range)` each load waited for its own round trip)
#pragma unroll
        for (int h = 0; h < 10 / MB; h++) {
            double2 s1[MB], s2[MB];
            int c1[MB], c2[MB];
#pragma unroll
            for (int u = 0; u < MB; u++) {
                const int t = MB * h + u, m = lane + 64 * t;
                const bool on = m <= 600 && (FULL ? (m & 1) == 0 : m % q == 0);
                const int mq = FULL ? (m >> 1) : m / q;
                const int b1 = mq - b_lo, b2 = (half - mq) - b_lo;
                c1[u] = (on && b1 >= 0 && b1 < a.nb) ? b1 : -1;
                c2[u] = (on && b2 >= 0 && b2 < a.nb && b2 != b1) ? b2 : -1;
                s1[u] = *reinterpret_cast<const double2 *>(a.slopes + 2 * (c1[u] < 0 ? 0 : c1[u]));
                s2[u] = *reinterpret_cast<const double2 *>(a.slopes + 2 * (c2[u] < 0 ? 0 : c2[u]));
            }
#pragma unroll
            for (int u = 0; u < MB; u++) {
                const int t = MB * h + u;
                if (c1[u] >= 0) {
                    pu[c1[u]] = fabs(pk[t]) * s1[u].x;       // norm_sqr * multiplier (src/spectrum.rs:426-428)
                    pd[c1[u]] = fabs(sqrt(pk[t])) * s1[u].y; // norm * multiplier (:432-434)
                }
                if (c2[u] >= 0) {
                    pu[c2[u]] = fabs(pn[t]) * s2[u].x;
                    pd[c2[u]] = fabs(sqrt(pn[t])) * s2[u].y;
                }
            }
        }
        wave_sync();
        VBX_PHASE(a.work, f, 13);
        if (a.num_coeffs <= 16) mfcc_tail_q(pu, pd, en, a.bins, a.dct, a.num_coeffs, b_lo, lane, a.out_mfcc + f * a.mfcc_ld, a.work, f, (a.pcm & SP_FLAG_MFCC_DEFER) != 0);
        else mfcc_tail_m(pu, pd, en, a.bins, a.dct, a.num_coeffs, b_lo, lane, a.out_mfcc + f * a.mfcc_ld);
        if (a.mfcc_status != nullptr && lane == 0) a.mfcc_status[f] = 0;
        wave_sync();
    }

    VBX_PHASE(a.work, f, 3);
    if constexpr (!PITCH) return;

    // ---- second transform: Y = FFT(G);  S[2j] = Re Y[j] / M, S[2j+1] = -Im Y[j] / M, j < 600 only ----
    fft1200<2, (WAVES >= 3)>(re, im, xr, xi, ex, a.tab, t2);
    VBX_PHASE(a.work, f, 4);

    // r[lag] = (S[lag] - x0 x[lag]) + x0 (Q1), lane l: j = l + 64 t (kc = 0) and j = 400 + l + 64 t < 600 (kc = 1)
    constexpr double INV_M = 1.0 / (double)SP_M;
    double r_e[11], r_o[11];                                 // slots 0..6: kc = 0, t = 0..6;  7..10: kc = 1, t = 0..3
    int jj[11];
#pragma unroll
    for (int s = 0; s < 11; s++) {
        const int t = (s < 7) ? s : s - 7, kc = (s < 7) ? 0 : 1;
        const int q = lane + 64 * t;
        const bool ok = (kc == 0) ? (q < 400) : (q < 200);
        jj[s] = ok ? q + 400 * kc : -1;
        r_e[s] = xr[t][kc] * INV_M;
        r_o[s] = -(xi[t][kc] * INV_M);
    }
    const double s0 = readlane_f64(r_e[0], 0);               // S[0], the scale of the transform's rounding error
    if (x0 != 0.0) {                                         // rectangular frames: the fold seed differs from S (uniform branch)
#pragma unroll
        for (int s = 0; s < 11; s++) {
            const int i = 2 * jj[s];
            if (jj[s] >= 0 && i < n) {
                const double xs = (FULL && (a.pcm & SP_FLAG_PCM)) ? pcm16_value(x16[i]) : xf[i];
                const double xe = (a.window != nullptr) ? xs * a.window[i] : xs;
                r_e[s] = (r_e[s] - x0 * xe) + x0;
            }
            if (jj[s] >= 0 && i + 1 < n) {
                const double xs = (FULL && (a.pcm & SP_FLAG_PCM)) ? pcm16_value(x16[i + 1]) : xf[i + 1];
                const double xo = (a.window != nullptr) ? xs * a.window[i + 1] : xs;
                r_o[s] = (r_o[s] - x0 * xo) + x0;
            }
        }
    }
    if constexpr (MODE == SP_AC_ONLY) {                      // autocorrelate(n_lags): the lag sums and nothing else
        double *row = a.out_r + f * (long)a.n_lags;
        const bool al = ((((uintptr_t)a.out_r) & 15) == 0) && (a.n_lags & 1) == 0;      // uniform: every row 16-byte aligned
#pragma unroll
        for (int s = 0; s < 11; s++) {
            const int i = 2 * jj[s];
            if (jj[s] >= 0 && i + 1 < a.n_lags && al) *reinterpret_cast<double2 *>(row + i) = double2{r_e[s], r_o[s]};
            else {
                if (jj[s] >= 0 && i < a.n_lags) row[i] = r_e[s];
                if (jj[s] >= 0 && i + 1 < a.n_lags) row[i + 1] = r_o[s];
            }
        }
        return;
    }
    if (LPC) {                                               // the raw autocorrelation r[0..12] into the frame's LPC row: lane l holds r[2l], r[2l + 1];
        // levinson_rows_kernel_t makes it LPC::lpc(12) afterwards, one row per lane (vbx_spectral.hpp)
        static_assert(SP_LPC_P == 12, "seven lanes hold r[0..12]");
        double *row = a.out_lpc + f * a.lpc_ld;
        if (lane < 7) { row[2 * lane] = r_e[0]; if (lane < 6) row[2 * lane + 1] = r_o[0]; }
    }
    double amax = -1.0;                                      // max_amplitude over ALL n lags (Q2; NaN never wins)
#pragma unroll
    for (int s = 0; s < 11; s++) {
        const int i = 2 * jj[s];
        const double ae = fabs(r_e[s]), ao = fabs(r_o[s]);
        if (jj[s] >= 0 && i < n) amax = (ae > amax) ? ae : amax;
        if (jj[s] >= 0 && i + 1 < n) amax = (ao > amax) ? ao : amax;
    }
    amax = wave_max(amax);
    const double scale = 1.0 / amax;                         // normalize (:404), then / lag window (:406-408)
    double *ys = smem;
    wave_sync();                                             // every lane is done with the exchange buffer
    // uniform: the table's reciprocals serve (quotient_by_table, vbx_spectral.hpp) unless the scale is not a normal finite number
    const bool by_table = (a.pcm & SP_FLAG_LAG_RCP) != 0 && fabs(scale) < 1e290 && fabs(scale) > 1e-290;
    const double *lag_rcp = a.lag_window + lag_rcp_offset(n);
    if (by_table) {
        // (window entries and reciprocals of a few slots requested together, without a condition -- entry 0 stands in for a slot
        // without lags --, then their quotients: all eleven slots' pairs at once are 88 registers the instance does not have)
#ifndef VBX_EXP_LB
#define VBX_EXP_LB 4
#endif
        constexpr int LB = VBX_EXP_LB;
#pragma unroll
        for (int h = 0; h < (11 + LB - 1) / LB; h++) {
            double2 lwv[LB], rwv[LB];
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int s = LB * h + u;
                if (s >= 11) continue;
                const int i = 2 * jj[s];
                const int at = (jj[s] >= 0 && i + 1 < n) ? i : 0;
                lwv[u] = *reinterpret_cast<const double2 *>(a.lag_window + at);
                rwv[u] = *reinterpret_cast<const double2 *>(lag_rcp + at);
            }
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int s = LB * h + u;
                if (s >= 11) continue;
                const int i = 2 * jj[s];
                if (jj[s] >= 0 && i + 1 < n) {
                    double2 y;
                    y.x = quotient_by_table(r_e[s] * scale, lwv[u].x, rwv[u].x);
                    y.y = quotient_by_table(r_o[s] * scale, lwv[u].y, rwv[u].y);
                    *reinterpret_cast<double2 *>(ys + i) = y;
                } else if (jj[s] >= 0 && i < n) {            // the last lag of an odd n
                    ys[i] = (r_e[s] * scale) / a.lag_window[i];
                }
            }
            asm volatile("" ::: "memory");                   // the next batch's loads stay behind this one's quotients (registers)
        }
    } else {
        // (the lag window's entries requested together, without a condition: entry 0 stands in for a slot without lags)
        double2 lwv[11];
#pragma unroll
        for (int s = 0; s < 11; s++) {
            const int i = 2 * jj[s];
            lwv[s] = *reinterpret_cast<const double2 *>(a.lag_window + ((jj[s] >= 0 && i + 1 < n) ? i : 0));
        }
#pragma unroll
        for (int s = 0; s < 11; s++) {
            const int i = 2 * jj[s];
            if (jj[s] >= 0 && i + 1 < n) {
                const double2 lw = lwv[s];
                double2 y;
                y.x = (r_e[s] * scale) / lw.x;
                y.y = (r_o[s] * scale) / lw.y;
                *reinterpret_cast<double2 *>(ys + i) = y;
            } else if (jj[s] >= 0 && i < n) {                // the last lag of an odd n
                ys[i] = (r_e[s] * scale) / a.lag_window[i];
            }
        }
    }
    if (lane < Y_PAD) ys[n + lane] = 0.0;
#ifndef VBX_EXP_NO_EXACT_TAIL
    if (!FULL) spectral_exact_tail(ys, n, xf, a.window, a.lag_window, x0, scale, lane);
#endif
    wave_sync();
    VBX_PHASE(a.work, f, 5);
    // Rounding error of the two transforms: a few ulp of S[0] per lag (measured: < 8 eps S[0]); y = r * scale / w_lag
    // with w_lag >= 1/6 on the searched half.  SP_UNC_EPS bounds the error of a DIFFERENCE of two entries with a wide
    // margin; frames with a peak decision inside it go to the direct-sum kernel (launch_pitch_list).
    const double unc_tol = SP_UNC_EPS * fabs(s0) * scale;
    double2 *full = a.pp.full_off > 0 ? reinterpret_cast<double2 *>(reinterpret_cast<char *>(smem) + a.pp.full_off)
                  : a.pp.full_off < 0 ? reinterpret_cast<double2 *>(a.out_cand + f * a.cand_ld) : nullptr;
    if (!pitch_refine_store(ys, n, a.pp, f, a.out_cand, a.cand_ld, a.out_count, a.pitch_status, a.work, unc_tol, full)) {
        if (lane == 0) a.unsure_list[atomicAdd(a.unsure_count, 1)] = (int32_t)f;
    }
}

static size_t spectral_lds_bytes(int n) {
    size_t need = (size_t)pitch_refine_lds_bytes(n);
    const size_t t2_end = (size_t)SP_T2_LDS_OFFSET + 60 * sizeof(double2);     // exchange buffer | mel sums | T2 copy
    if (t2_end > need) need = t2_end;
    return (need + 15) & ~(size_t)15;
}

int spectral_plan(int n) {
    static const int min_n = [] { const char *e = getenv("VBX_SPECTRAL_MIN_N"); return e ? atoi(e) : SPECTRAL_MIN_N; }();
    if (n < min_n || n < 64) return SPECTRAL_PLAN_NONE;
    if (n <= 1024) return SPECTRAL_PLAN_1024;
    if (n <= SP_N) return SPECTRAL_PLAN_1200;
    if (n <= 2048) return SPECTRAL_PLAN_2048;
    if (n <= 4096) return SPECTRAL_PLAN_4096;
    return SPECTRAL_PLAN_NONE;
}

int spectral_plan_nc(int plan) {
    return plan == SPECTRAL_PLAN_1200 ? SP_N : plan == SPECTRAL_PLAN_1024 ? 1024 : plan == SPECTRAL_PLAN_2048 ? 2048 :
           plan == SPECTRAL_PLAN_4096 ? 4096 : 0;
}

int spectral_tab_complex(int plan) {
    return plan == SPECTRAL_PLAN_1200 ? SPECTRAL_TAB_COMPLEX : spectral_pow2_tab_complex(plan);
}

// twiddles, evaluated in long double and rounded once
void spectral_fill_tab(int plan, double *h) {
    if (plan != SPECTRAL_PLAN_1200) { spectral_pow2_fill_tab(plan, h); return; }
    const long double two_pi = 6.283185307179586476925286766559005768L;
    size_t o = 0;
    auto put = [&](long num, long den) {             // e^{-2 pi i num / den}
        const long double ang = two_pi * (long double)(num % den) / (long double)den;
        h[o++] = (double)cosl(ang); h[o++] = (double)(-sinl(ang));
    };
    for (long np = 0; np < 60; np++) for (long ka = 0; ka < 20; ka++) put(np * ka, 1200);
    for (long c = 0; c < 3; c++) for (long kb = 0; kb < 20; kb++) put(c * kb, 60);
    for (long m = 0; m <= 600; m++) put(m, 2400);
}

// The fused kernel serves: pitch for every frame length with a plan; LPC of order 12 with it; MFCC with it when the frame
// divides the transform (the mel filters read the n-point DFT bins = every (M / n)-th bin of the M-point transform).
bool spectral_supported(int n, int lpc_order, int mfcc_nb, int mfcc_b_lo, int num_coeffs) {
    const int plan = spectral_plan(n);
    if (plan == SPECTRAL_PLAN_NONE) return false;
    if (lpc_order != 0 && lpc_order != SP_LPC_P) return false;
    if (num_coeffs != 0 && ((2 * spectral_plan_nc(plan)) % n != 0 || (n & 1) || num_coeffs > 64 || mfcc_nb < 1 || mfcc_b_lo < 0 || mfcc_b_lo + mfcc_nb > n / 2)) return false;
    return true;
}

// The plan for the fused frame loop when MFCC is wanted: the frame's DFT bins must be bins of the transform, i.e. the frame
// length must divide the transform's M = 2 Nc.  512 divides the 1024 plan's 2048; 600 and 800 do not, but they divide the
// 1200 plan's 2400 -- a 17 % longer transform instead of a second kernel over the same frames (MFCC::mfcc alone costs about
// what the whole fused kernel does).
int spectral_plan_mfcc(int n) {
    const int plan = spectral_plan(n);
    if (plan == SPECTRAL_PLAN_NONE) return plan;
    if ((2 * spectral_plan_nc(plan)) % n == 0) return plan;
    if (n <= SP_N && (2 * SP_N) % n == 0 && !(n & 1)) return SPECTRAL_PLAN_1200;
    return plan;
}
bool spectral_supported_plan(int plan, int n, int lpc_order, int mfcc_nb, int mfcc_b_lo, int num_coeffs) {
    if (plan == SPECTRAL_PLAN_NONE || n > spectral_plan_nc(plan)) return false;
    if (lpc_order != 0 && lpc_order != SP_LPC_P) return false;
    if (num_coeffs != 0 && ((2 * spectral_plan_nc(plan)) % n != 0 || (n & 1) || num_coeffs > 64 || mfcc_nb < 1 || mfcc_b_lo < 0 || mfcc_b_lo + mfcc_nb > n / 2)) return false;
    return true;
}

// ---- tables of the interpolated MFCC bins (mfcc_interp_t, vbx_kernels.hpp), host side, in long double ----
// threads per frame of the plan's kernel (the bins are dealt to them) and the LDS the interpolation may use: what costs at most one
// of the twelve / eight frames of a CU (a chirp-z kernel over the same frames costs more), none of the 4096-point plan's four
static void mfcc_interp_geom(int plan, long *M, int *nt, int *lds_cap) {
    *M = 2L * spectral_plan_nc(plan);
    *nt = plan == SPECTRAL_PLAN_4096 ? 128 : 64;
    *lds_cap = (plan == SPECTRAL_PLAN_1200 || plan == SPECTRAL_PLAN_1024) ? (160 * 1024) / 11 : plan == SPECTRAL_PLAN_2048 ? (160 * 1024) / 7 : (160 * 1024) / 4;
}
static int mfcc_interp_slots(int plan, int nb) { long M; int nt, cap; mfcc_interp_geom(plan, &M, &nt, &cap); return (nb + nt - 1) / nt; }

// Taps per bin.  The cut's error falls like e^{-pi tau W} / (pi W / 2) (tau = 1/2 - n / 2M, the Kaiser-Bessel bump's half-width; the sinc's
// envelope at the cut): 40 taps up to M / n = 2.47, 32 up to 5, 24 beyond hold it at ~1e-15 of the transform's largest |X| -- the size of
// the transform's own rounding, so that the interpolation is never the limiting term, whatever the frame's dynamic range
// (tests/test_mfcc_interp_table.py measures every class against the exact DFT: < 1e-14; tests/test_gpu_analyze.py: a pure tone, whose
// filters hold only leakage).  The first choice of round 5 -- 32 taps at M / n = 2.18: 4e-13 -- measured no faster (29.3 against 29.6 M frames/s
// at 1103 / 441 on one box, 30.3 against 30.0 on another: ~1 %) and passed the same tests; kept as VBX_EXP_INTERP_FEWER_TAPS.
int mfcc_interp_taps(int plan, int n) {
    long M; int nt, cap; mfcc_interp_geom(plan, &M, &nt, &cap);
    const double tau = 0.5 - (double)n / (2.0 * (double)M);
#ifdef VBX_EXP_INTERP_FEWER_TAPS
    return tau >= 0.36 ? 24 : tau >= 0.268 ? 32 : MFCC_INTERP_MAX_TAPS;
#endif
    return tau >= 0.401 ? 24 : tau >= 0.298 ? 32 : MFCC_INTERP_MAX_TAPS;     // the smallest W of 24 / 32 / 40 with e^{-pi tau W} / (pi W / 2) <= 2e-15
}

size_t mfcc_interp_table_bytes(int plan, int nb) {
    long M; int nt, cap; mfcc_interp_geom(plan, &M, &nt, &cap);
    const size_t slots = (size_t)mfcc_interp_slots(plan, nb);
    return (size_t)(M / 4 + 1) * 16 + slots * (MFCC_INTERP_MAX_TAPS / 2) * nt * 16 + slots * nt * 4;
}
size_t mfcc_interp_coef_offset(int plan) { return (size_t)(2L * spectral_plan_nc(plan) / 4 + 1) * 16; }
size_t mfcc_interp_j0_offset(int plan, int nb) {
    long M; int nt, cap; mfcc_interp_geom(plan, &M, &nt, &cap);
    return mfcc_interp_coef_offset(plan) + (size_t)mfcc_interp_slots(plan, nb) * (MFCC_INTERP_MAX_TAPS / 2) * nt * 16;
}

bool mfcc_interp_fill(int plan, int n, int b_lo, int nb, void *h_table, mfcc_interp_t *d) {
    if (plan == SPECTRAL_PLAN_NONE) return false;
    long M; int NT, cap;
    mfcc_interp_geom(plan, &M, &NT, &cap);
    const long quarter = M / 4;
    if (n < 2 || 2L * n > M || M % n == 0 || nb < 1 || nb > 4096 || b_lo < 0) return false;
    const int SL = mfcc_interp_slots(plan, nb);
    const int W = mfcc_interp_taps(plan, n), HT = W / 2;
    const long double pi = 3.141592653589793238462643383279502884L;
    const long double tau = 0.5L - (long double)n / (2.0L * (long double)M);                   // the bump's half-width
    const long double beta = pi * tau * (long double)W;
    auto sinhc = [](long double s) { return s < 1e-6L ? 1.0L + s * s / 6.0L : sinhl(s) / s; };
    const long double norm = sinhc(beta);
    char *base = static_cast<char *>(h_table);
    double *rot = reinterpret_cast<double *>(base);
    double *coef = reinterpret_cast<double *>(base + mfcc_interp_coef_offset(plan));
    int32_t *j0t = reinterpret_cast<int32_t *>(base + mfcc_interp_j0_offset(plan, nb));
    for (long j = 0; j <= quarter; j++) {                                                      // e^{2 pi i j c / M}, c = (n - 1) / 2
        const long double ang = 2.0L * pi * (long double)((j * (long)(n - 1)) % (2 * M)) / (long double)(2 * M);
        rot[2 * j] = (double)cosl(ang); rot[2 * j + 1] = (double)sinl(ang);
    }
    long jmin = 1L << 40, jmax = -(1L << 40);
    for (int i = 0; i < nb; i++) {
        const long first = ((long)(b_lo + i) * M) / n - W / 2 + 1;
        if (first < jmin) jmin = first;
        if (first + W - 1 > jmax) jmax = first + W - 1;
    }
    if (jmax > quarter || jmin < -quarter) return false;
    for (int i = 0; i < NT * SL; i++) {
        const int u = i / NT, th = i % NT;
        const long k = b_lo + i;
        const long first = (i < nb) ? (k * M) / n - W / 2 + 1 : jmin;
        j0t[i] = (int32_t)(first - jmin);
        for (int t = 0; t < W; t++) {
            long double c = 0.0L;
            if (i < nb) {
                const long num = k * M - (first + t) * (long)n;                                // nu = num / n, |nu| <= W / 2
                if (num == 0) c = 1.0L;
                else {
                    const long double nu = (long double)num / (long double)n;
                    long r = num % (2L * n); if (r < 0) r += 2L * n;                           // sin(pi nu) from the reduced numerator
                    const long double sn = sinl(pi * (long double)r / (long double)n);
                    const long double arg = beta * beta - (2.0L * pi * tau * nu) * (2.0L * pi * tau * nu);
                    const long double bump = sinhc(arg > 0.0L ? sqrtl(arg) : 0.0L) / norm;
                    c = sn / (pi * nu) * bump;
                }
            }
            coef[(((size_t)u * HT + t / 2) * NT + th) * 2 + (t & 1)] = (double)c;
        }
    }
    const int zn = (int)(jmax - jmin + 1), nbp = (nb + 1) & ~1;
    d->jmin = (int)jmin; d->jmax = (int)jmax; d->taps = W;
    d->pu_off = 2 * zn;
    d->lds_bytes = (2 * zn + 2 * nbp + 64) * 8;
    d->rot = nullptr; d->coef = nullptr; d->j0 = nullptr;                                      // the caller's: device addresses
    return d->lds_bytes <= cap;
}

int launch_analyze(hipStream_t s, const spectral_launch_t &L) {
    spectral_args_t a;
    a.frames = L.x; a.n_frames = L.F; a.stride = L.stride; a.window = L.window; a.lag_window = L.lag_window;
    a.tab = reinterpret_cast<const double2 *>(L.tab);
    a.n = L.n;
    a.pp.sample_rate = L.sample_rate; a.pp.threshold = L.threshold; a.pp.fmin = L.fmin; a.pp.fmax = L.fmax; a.pp.kmax = L.kmax; a.pp.f32 = 0;
    a.out_cand = reinterpret_cast<double *>(L.out_cand); a.cand_ld = L.cand_ld; a.out_count = L.out_count;
    a.pitch_status = L.pitch_status; a.work = L.work;
    a.out_lpc = L.out_lpc; a.lpc_ld = L.lpc_ld;
    a.out_mfcc = L.out_mfcc; a.mfcc_ld = L.mfcc_ld; a.mfcc_status = L.mfcc_status;
    a.bins = L.bins; a.slopes = L.slopes; a.dct = L.dct; a.num_coeffs = L.num_coeffs; a.nb = L.nb;
    a.unsure_list = L.unsure_list; a.unsure_count = L.unsure_count;
    a.out_r = L.out_r; a.n_lags = L.n_lags;
    a.pcm = ((L.pcm && L.n == SP_N) ? SP_FLAG_PCM : 0) | (L.lag_rcp ? SP_FLAG_LAG_RCP : 0) | ((L.mfcc_defer && L.num_coeffs <= 16) ? SP_FLAG_MFCC_DEFER : 0);      // (PCM: the host side only asks for it on full 1200-sample frames)
    a.mfcc_q = (L.plan != SPECTRAL_PLAN_NONE && L.n > 0) ? (2 * spectral_plan_nc(L.plan)) / L.n : 2;
    a.ip = mfcc_interp_t{};
    if (L.plan != SPECTRAL_PLAN_1200) return launch_analyze_pow2(s, L, a);
    const dim3 grid((unsigned)L.F), block(64);
    const size_t base = spectral_lds_bytes(L.n);
    size_t extra = pitch_full_list_bytes(L.n, L.kmax);
    a.pp.full_off = extra ? (int)base : 0;
    // kmax = VBX_PITCH_MAX_CANDIDATES(frame_len) (the whole Vec of every frame fits its output row): the refined candidates
    // are parked in the row itself instead of an extra LDS region, which keeps the frame state at 13.5 KB = twelve
    // wavefronts per CU (the form compiled for three wavefronts per SIMD, below)
    if (extra && L.kmax >= pitch_full_list_entries(L.n) && L.out_r == nullptr && !L.mfcc_only) { a.pp.full_off = -1; extra = 0; }
    const size_t lds = base + extra;
    const bool lpc = L.out_lpc != nullptr, mf = L.out_mfcc != nullptr;
    if (L.mfcc_only) {                                       // n == SP_N, or a padded frame with interpolated bins
        if (L.interp && L.n != SP_N) {
            a.ip = L.ip;
            const size_t li = spectral_lds_bytes(0) > (size_t)L.ip.lds_bytes ? spectral_lds_bytes(0) : (((size_t)L.ip.lds_bytes + 15) & ~(size_t)15);
            hipLaunchKernelGGL((analyze_kernel<false, true, false, SP_MFCC_ONLY_INTERP>), grid, block, li, s, a);
        } else hipLaunchKernelGGL((analyze_kernel<false, true, true, SP_MFCC_ONLY>), grid, block, spectral_lds_bytes(0), s, a);
        return 0;
    }
    if (L.out_r != nullptr) {                                // autocorrelate(n_lags) alone
        if (L.n == SP_N) hipLaunchKernelGGL((analyze_kernel<false, false, true, SP_AC_ONLY>), grid, block, spectral_lds_bytes(0), s, a);
        else hipLaunchKernelGGL((analyze_kernel<false, false, false, SP_AC_ONLY>), grid, block, spectral_lds_bytes(0), s, a);
        return 0;
    }
    // Three wavefronts per SIMD (twelve frames of 13.5 KB fill the CU's LDS; 168 registers) wherever the frame state allows it,
    // i.e. no full-list region in LDS.  Round 3 had it for pitch alone from kmax = 2 (the refinement is a chain of dependent
    // operations that two wavefronts do not cover); at kmax = 1 and in the fused loop the transforms' ~55 spilled registers
    // cost more than the third wavefront brought.  Round 4: with the twiddle products in pinned batches (twiddle_tight) the
    // 168-register instances spill 12-29 registers instead of 146-171, and three wavefronts win everywhere: pitch at kmax = 1
    // 33.7 -> 37.8 M frames/s, the fused loop (the headline) 30.0 -> 33.5 M.  VBX_SPECTRAL_W3=0: two wavefronts as before (A/B).
    static const bool want3 = [] { const char *e = getenv("VBX_SPECTRAL_W3"); return e == nullptr || atoi(e) != 0; }();
    const bool w3 = want3 && extra == 0 && 12 * lds <= 160 * 1024;
#define VBX_SP_LAUNCH(LPC_, MF_, FULL_)                                                                              \
    do {                                                                                                              \
        if (w3) hipLaunchKernelGGL((analyze_kernel<LPC_, MF_, FULL_, SP_ANALYZE, 3>), grid, block, lds, s, a);        \
        else hipLaunchKernelGGL((analyze_kernel<LPC_, MF_, FULL_>), grid, block, lds, s, a);                          \
    } while (0)
    if (L.interp && mf && L.n != SP_N) {                     // a padded frame whose MFCC bins are interpolated from the transform's
        a.ip = L.ip;
        const size_t li = lds > (size_t)L.ip.lds_bytes ? lds : (((size_t)L.ip.lds_bytes + 15) & ~(size_t)15);
        const bool w3i = want3 && extra == 0 && 12 * li <= 160 * 1024;
        if (lpc) { if (w3i) hipLaunchKernelGGL((analyze_kernel<true, true, false, SP_ANALYZE_INTERP, 3>), grid, block, li, s, a);
                   else hipLaunchKernelGGL((analyze_kernel<true, true, false, SP_ANALYZE_INTERP>), grid, block, li, s, a); }
        else { if (w3i) hipLaunchKernelGGL((analyze_kernel<false, true, false, SP_ANALYZE_INTERP, 3>), grid, block, li, s, a);
               else hipLaunchKernelGGL((analyze_kernel<false, true, false, SP_ANALYZE_INTERP>), grid, block, li, s, a); }
    } else if (L.n != SP_N) {                                // a padded frame; MFCC from the transform's own bins when its length divides 2400
        if (lpc && mf) VBX_SP_LAUNCH(true, true, false);
        else if (mf) VBX_SP_LAUNCH(false, true, false);
        else if (lpc) VBX_SP_LAUNCH(true, false, false);
        else VBX_SP_LAUNCH(false, false, false);
    } else if (lpc && mf) VBX_SP_LAUNCH(true, true, true);
    else if (lpc) VBX_SP_LAUNCH(true, false, true);
    else if (mf) VBX_SP_LAUNCH(false, true, true);
    else VBX_SP_LAUNCH(false, false, true);
#undef VBX_SP_LAUNCH
    return 0;
}

}  // namespace vbx
