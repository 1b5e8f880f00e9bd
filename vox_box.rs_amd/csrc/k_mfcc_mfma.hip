// k_mfcc_mfma.hip -- MFCC::mfcc (src/spectrum.rs:401-441, Q14) with both DFT stages on the FP64 matrix cores.
//
// Same two-stage DFT of the needed bins as mfcc_dft2_kernel (k_mfcc.hip): n = n2*i1 + i2, k = k1 + n1*k2,
//   A[i2][k1] = sum_i1 x[n2*i1 + i2] * W_n1^(i1*k1)                          stage 1
//   X[k]      = sum_i2 (A[i2][k1] * W_n^(i2*k1)) * W_n2^(i2*k2)               twiddle, stage 2
// but as v_mfma_f64_16x16x4_f64 products, one wavefront per frame:
//   stage 1   D1[i2][c] = sum_i1 Xp[i1][i2] * C[i1][c]       M = i2 (MT tiles), N = cos columns k1 = 0..16*NTD-1 then the
//             sin columns of the same k1 (2*NTD tiles), K = i1.  A operand from the frame in LDS, B from a table in LDS.
//   twiddle   in registers.  The C/D layout of an accumulator (lane l, register r <-> row (l>>4)+4r, column l&15) IS the
//             B-operand layout of K-step 4*mt + r (k = l>>4, column l&15): register r of the cos tile and of the sin tile
//             of the same lane hold Re and -Im of A[i2][k1]; one complex multiply by a table entry gives B[i2][k1], and
//             with the conjugate the mirrored column B[i2][n1-k1] (real input), ready to be fed back -- the
//             intermediate never touches LDS.
//   stage 2   D2[2*k2+p][k1] = sum_kk Wm[2*k2+p][kk] * Bm[kk][k1]   kk = (Re rows i2, then Im rows i2): rows 2*k2 / 2*k2+1
//             are Re / Im of X[k1 + n1*k2].  The A operand (Wm, the same for every frame) lives in registers.
// At n = 1200 (n1 = 30, n2 = 40): 48 + 48 MFMA per frame.  Then |X|^2, |X|, the mel sums and the DCT as in k_mfcc.hip.
// Plans that do not fit (k2 > 8 rows, n1 > 63, n2 > 64) use the vector kernels of k_mfcc.hip.
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"
#include "vbx_mfcc_tail.hpp"

namespace vbx {

typedef double mf_d4 __attribute__((ext_vector_type(4)));

// LDS: per block  ctab[n1p][32*NTD] | twd[MT*NTD*4][64][2] | twm[MT*NTM*4][64][2]
//      per wave   xp[n1p][n2p] | X[2][nbp] | en[64]                        (n1p = 4*ceil(n1/4), n2p = 16*MT)
template <int MT, int NTD, int NTM>
__global__ __launch_bounds__(512) void mfcc_mfma_kernel(
    const double *__restrict__ x, long n_frames, int n, long stride, const double *__restrict__ window,
    const double *__restrict__ ctab_g, const double *__restrict__ twd_g, const double *__restrict__ twm_g,
    const double *__restrict__ wm_g /* [8*MT][64] */, int n1, int n2, int k2n, int mir_src0, int mir_src1,
    const int32_t *__restrict__ bins, const double *__restrict__ slopes, const double *__restrict__ dct_table,
    int num_coeffs, int nb, double *__restrict__ out, long out_ld, int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NC = 32 * NTD;                         // stage-1 table columns: cos tiles, then sin tiles
    constexpr int N2P = 16 * MT;
    const int wave = threadIdx.x >> 6, lane = lane_id(), W = blockDim.x >> 6;
    const int n1p = (n1 + 3) & ~3, ks1 = n1p >> 2;
    const int nbp = (nb + 1) & ~1;
    double *ct = smem;
    double *twd = ct + (size_t)n1p * NC;
    double *twm = twd + (size_t)MT * NTD * 4 * 128;
    const size_t per_wave = (size_t)n1p * N2P + 2 * (size_t)nbp + 64;
    double *xp = twm + (size_t)MT * NTM * 4 * 128 + (size_t)wave * per_wave;
    double *xre = xp + (size_t)n1p * N2P, *xim = xre + nbp, *en = xim + nbp;
    for (int i = threadIdx.x; i < n1p * NC; i += blockDim.x) ct[i] = ctab_g[i];
    for (int i = threadIdx.x; i < MT * NTD * 4 * 128; i += blockDim.x) twd[i] = twd_g[i];
    for (int i = threadIdx.x; i < MT * NTM * 4 * 128; i += blockDim.x) twm[i] = twm_g[i];
    for (int i = lane; i < n1p * N2P; i += 64) xp[i] = 0.0;           // the pad cells stay zero for every frame
    __syncthreads();

    const int row = lane & 15, kq = lane >> 4;           // A operand: (row, k); B operand and C/D: (k or row group, column)
    const int b_lo = bins[0], top = b_lo + nb;
    double wreg[8 * MT];                                 // stage-2 A operand of this lane, all K-steps
#pragma unroll
    for (int s = 0; s < 8 * MT; s++) wreg[s] = wm_g[s * 64 + lane];
    // sample i of the frame lives at xp[(i / n2) * N2P + i % n2]; this lane owns i = lane + 64 j
    constexpr int PF = 20;
    int poff[PF];
    double wwin[PF], pre[PF];
    const long fstep = (long)gridDim.x * W;
    long f = (long)blockIdx.x * W + wave;
#pragma unroll
    for (int j = 0; j < PF; j++) {
        const int i = 64 * j + lane;
        poff[j] = (i / n2) * N2P + (i % n2);
        wwin[j] = (window != nullptr && i < n) ? window[i] : 1.0;
        pre[j] = (f < n_frames && i < n) ? x[f * stride + i] : 0.0;
    }
    for (; f < n_frames; f += fstep) {
        const double *xf = x + f * stride;
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const int i = 64 * j + lane;
            if (i < n) xp[poff[j]] = pre[j] * wwin[j];
        }
        for (int i = 64 * PF + lane; i < n; i += 64)
            xp[(i / n2) * N2P + (i % n2)] = (window != nullptr) ? xf[i] * window[i] : xf[i];
        if (f + fstep < n_frames) {
            const double *xn = xf + fstep * stride;
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const int i = 64 * j + lane;
                pre[j] = (i < n) ? xn[i] : 0.0;
            }
        }
        wave_sync();

        // stage 1
        mf_d4 acc1[MT][2 * NTD];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < 2 * NTD; nt++) acc1[mt][nt] = mf_d4{0.0, 0.0, 0.0, 0.0};
        {
            const double *ap = xp + kq * N2P + row;      // A[row][k] = xp[4s + k][16 mt + row]
            const double *bp = ct + kq * NC + row;       // B[k][col] = ct[4s + k][16 nt + col]
            for (int s = 0; s < ks1; s++) {
                double bv[2 * NTD];
#pragma unroll
                for (int nt = 0; nt < 2 * NTD; nt++) bv[nt] = bp[16 * nt];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const double av = ap[16 * mt];
#pragma unroll
                    for (int nt = 0; nt < 2 * NTD; nt++)
                        acc1[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[nt], acc1[mt][nt], 0, 0, 0);
                }
                ap += 4 * N2P; bp += 4 * NC;
            }
        }
        // twiddle + stage 2, one column tile at a time; results to LDS as Re/Im rows indexed by bin
#pragma unroll
        for (int jt = 0; jt < NTD + NTM; jt++) {
            const bool mirror = jt >= NTD;
            const int src = mirror ? ((jt - NTD == 0) ? mir_src0 : mir_src1) : jt;     // direct tile the columns come from
            mf_d4 acc2 = mf_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double ac, as;
                    // src is wave-uniform; NTD <= 2
                    if (NTD == 1 || src == 0) { ac = acc1[mt][0][r]; as = acc1[mt][NTD][r]; }
                    else { ac = acc1[mt][NTD - 1][r]; as = acc1[mt][2 * NTD - 1][r]; }
                    const double *tp = mirror ? twm + ((size_t)((mt * NTM + (jt - NTD)) * 4 + r) * 64 + lane) * 2
                                              : twd + ((size_t)((mt * NTD + jt) * 4 + r) * 64 + lane) * 2;
                    const double2 w = *reinterpret_cast<const double2 *>(tp);            // cos, sin of 2 pi i2 k1 / n
                    double bre, bim;
                    if (!mirror) { bre = fma(ac, w.x, -(as * w.y)); bim = -fma(as, w.x, ac * w.y); }   // (ac - i as)(cos - i sin)
                    else { bre = fma(ac, w.x, as * w.y); bim = fma(as, w.x, -(ac * w.y)); }            // (ac + i as)(cos - i sin)
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(wreg[4 * mt + r], bre, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(wreg[4 * MT + 4 * mt + r], bim, acc2, 0, 0, 0);
                }
            }
            const int kp = 16 * src + row;                                   // column of the source tile (row == l & 15)
            const int k1 = mirror ? n1 - kp : kp;
            const bool col_ok = mirror ? (kp >= 1 && k1 >= 16 * NTD && k1 < n1) : (k1 < n1);
#pragma unroll
            for (int r2 = 0; r2 < 4; r2++) {
                const int rowm = kq + 4 * r2;
                const int k2 = rowm >> 1;
                const int bin = k1 + n1 * k2;
                if (col_ok && k2 < k2n && bin >= b_lo && bin < top) {
                    if (rowm & 1) xim[bin - b_lo] = acc2[r2]; else xre[bin - b_lo] = acc2[r2];
                }
            }
        }
        wave_sync();
        for (int bi = lane; bi < nb; bi += 64) {
            const double re = xre[bi], im = xim[bi];
            const double m2 = fma(re, re, im * im);
            const double2 sl = *reinterpret_cast<const double2 *>(slopes + 2 * bi);
            xre[bi] = fabs(m2) * sl.x;                                       // norm_sqr * multiplier (:426-428)
            xim[bi] = fabs(sqrt(m2)) * sl.y;                                 // norm * multiplier (:432-434)
        }
        wave_sync();
        mfcc_tail_m(xre, xim, en, bins, dct_table, num_coeffs, b_lo, lane, out + f * out_ld);
        if (status != nullptr && lane == 0) status[f] = 0;
        wave_sync();
    }
}

// ---- plan ---------------------------------------------------------------------------------------------------------
static void mplan_tiles(int n1, int &ntd, int &ntm, int &s0, int &s1) {
    s0 = 0; s1 = 0;
    if (n1 <= 16) { ntd = 1; ntm = 0; }
    else if (n1 <= 31) { ntd = 1; ntm = 1; s0 = 0; }
    else if (n1 == 32) { ntd = 2; ntm = 0; }
    else if (n1 <= 47) { ntd = 2; ntm = 1; s0 = 0; }
    else { ntd = 2; ntm = 2; s0 = 0; s1 = 1; }
}

size_t mfcc_mfma_lds(const mfcc_mplan_t &pl, int nb, int waves) {
    const int n1p = (pl.n1 + 3) & ~3, nbp = (nb + 1) & ~1;
    const size_t shared = (size_t)n1p * 32 * pl.ntd + (size_t)pl.mt * (pl.ntd + pl.ntm) * 4 * 128;
    const size_t per_wave = (size_t)n1p * 16 * pl.mt + 2 * (size_t)nbp + 64;
    return (shared + (size_t)waves * per_wave) * sizeof(double);
}

// n1*n2 = n with the fewest MFMA per frame among the factorisations the kernel supports
mfcc_mplan_t mfcc_mfma_plan(int n, int b_lo, int nb) {
    mfcc_mplan_t best{};
    best.ok = false;
    long best_cost = 0;
    const int top = b_lo + nb;
    for (int n1 = 4; n1 <= 63 && n1 <= n / 2; n1++) {
        if (n % n1) continue;
        const int n2 = n / n1;
        if (n2 > 64 || n2 < 2) continue;
        const int k2 = (top + n1 - 1) / n1;
        if (k2 > 8 || k2 < 1) continue;
        mfcc_mplan_t pl{};
        pl.ok = true; pl.n1 = n1; pl.n2 = n2; pl.k2 = k2; pl.mt = (n2 + 15) / 16;
        mplan_tiles(n1, pl.ntd, pl.ntm, pl.src0, pl.src1);
        const long cost = (long)pl.mt * 2 * pl.ntd * ((n1 + 3) / 4) + 8L * pl.mt * (pl.ntd + pl.ntm);
        if (mfcc_mfma_lds(pl, nb, 1) > 160 * 1024) continue;
        if (!best.ok || cost < best_cost) { best = pl; best_cost = cost; }
    }
    return best;
}

template <int MT, int NTD, int NTM>
static void launch_mm(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                      const mfcc_mplan_t &pl, const double *ctab, const double *twd, const double *twm, const double *wm,
                      const int32_t *bins_dev, const double *slopes, const double *dct_table, int num_coeffs, double *out,
                      long out_ld, int32_t *status, int nb, int cu_count) {
    int w = 8;
    while (w > 1 && mfcc_mfma_lds(pl, nb, w) > 160 * 1024) w--;
    long blocks = (F + w - 1) / w;
    const long cap = (long)(cu_count > 0 ? cu_count : 256) * 8;        // grid-stride: tables are loaded once per block
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((mfcc_mfma_kernel<MT, NTD, NTM>), dim3((unsigned)blocks), dim3(64 * w), mfcc_mfma_lds(pl, nb, w), s,
                       x, F, n, stride, window, ctab, twd, twm, wm, pl.n1, pl.n2, pl.k2, pl.src0, pl.src1, bins_dev, slopes,
                       dct_table, num_coeffs, nb, out, out_ld, status);
}

template <int MT>
static void launch_mm_mt(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                         const mfcc_mplan_t &pl, const double *ctab, const double *twd, const double *twm, const double *wm,
                         const int32_t *bins_dev, const double *slopes, const double *dct_table, int num_coeffs, double *out,
                         long out_ld, int32_t *status, int nb, int cu_count) {
#define VBX_MM(A, B) launch_mm<MT, A, B>(s, x, F, n, stride, window, pl, ctab, twd, twm, wm, bins_dev, slopes, dct_table, num_coeffs, out, out_ld, status, nb, cu_count)
    if (pl.ntd == 1 && pl.ntm == 0) VBX_MM(1, 0);
    else if (pl.ntd == 1 && pl.ntm == 1) VBX_MM(1, 1);
    else if (pl.ntd == 2 && pl.ntm == 0) VBX_MM(2, 0);
    else if (pl.ntd == 2 && pl.ntm == 1) VBX_MM(2, 1);
    else VBX_MM(2, 2);
#undef VBX_MM
}

void launch_mfcc_mfma(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                      const mfcc_mplan_t &pl, const double *ctab, const double *twd, const double *twm, const double *wm,
                      const int32_t *bins_dev, const double *slopes, const double *dct_table, int num_coeffs, double *out,
                      long out_ld, int32_t *status, int nb, int cu_count) {
#define VBX_MT(M) launch_mm_mt<M>(s, x, F, n, stride, window, pl, ctab, twd, twm, wm, bins_dev, slopes, dct_table, num_coeffs, out, out_ld, status, nb, cu_count)
    switch (pl.mt) { case 1: VBX_MT(1); break; case 2: VBX_MT(2); break; case 3: VBX_MT(3); break; default: VBX_MT(4); break; }
#undef VBX_MT
}

}  // namespace vbx
