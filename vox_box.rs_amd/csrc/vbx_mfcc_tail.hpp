// vbx_mfcc_tail.hpp -- the tail of MFCC::mfcc shared by the matrix-core MFCC kernel and the fused spectral kernel:
// mel filter sums in reference order, clamped log10, DCT (src/spectrum.rs:421-439, :391-397; Q14).
#pragma once

#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

// mel energies and dct of one frame: the tail of k_mfcc.hip (kept identical)
__device__ __forceinline__ void mfcc_tail_m(const double *pu, const double *pd, double *en, const int32_t *bins,
                                            const double *dct_table, int num_coeffs, int b_lo, int lane,
                                            double *out_row) {
    if (lane < num_coeffs) {                              // lane w <-> filter w
        const int w0 = bins[lane], w1 = bins[lane + 1], w2 = bins[lane + 2];
        double up_sum = 0.0, down_sum = 0.0;
        for (int b = w0 - b_lo; b < w1 - b_lo; b += 4) {
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (b + j < w1 - b_lo) ? pu[b + j] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) up_sum = up_sum + v[j];
        }
        for (int b = w1 - b_lo; b < w2 - b_lo; b += 4) {
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (b + j < w2 - b_lo) ? pd[b + j] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) down_sum = down_sum + v[j];
        }
        const double lg = log10(up_sum + down_sum);
        en[lane] = (lg != lg || lg < 1.0e-10) ? 1.0e-10 : lg;   // f64::max(1e-10): NaN yields the other operand
    }
    wave_sync();
    if (lane < num_coeffs) {                              // dct (:391-397)
        double acc = 0.0;
        for (int j = 0; j < num_coeffs; j++) acc = acc + en[j] * dct_table[lane * num_coeffs + j];
        out_row[lane] = 2.0 * acc;
    }
}

// The same tail with FOUR lanes per filter (num_coeffs <= 16): a filter's bins are dealt round-robin to its quad and the
// partial sums meet in a quad reduction -- the widest filter spans ~80 bins, a serial chain of that length per frame
// otherwise.  The association of the sums differs from the reference's left-to-right fold (a few ulp, tolerance 1e-6).
// Must be called from converged code.
__device__ __forceinline__ void mfcc_tail_q(const double *pu, const double *pd, double *en, const int32_t *bins,
                                            const double *dct_table, int num_coeffs, int b_lo, int lane,
                                            double *out_row, unsigned long long *work = nullptr, long f = 0,
                                            const bool defer = false) {
    VBX_PHASE_INIT();
    const int w = lane >> 2, sub = lane & 3;
    const bool have = w < num_coeffs;
    double up_sum = 0.0, down_sum = 0.0;
    // Round 5: the same sums in the same order with the loads of four steps (and of the whole DCT row) requested together --
    // one element per trip, each addition waited for its own LDS read (20 trips for the widest filter), each DCT term for its own
    // load from memory.  A step past the filter's end adds +0.0 to a sum of non-negative terms: the same bits.
    if (have) {
        const int w0 = bins[w] - b_lo, w1 = bins[w + 1] - b_lo, w2 = bins[w + 2] - b_lo;
        for (int b = w0 + sub; b < w1; b += 16) {
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (b + 4 * j < w1) ? pu[b + 4 * j] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) up_sum += v[j];
        }
        for (int b = w1 + sub; b < w2; b += 16) {
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (b + 4 * j < w2) ? pd[b + 4 * j] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) down_sum += v[j];
        }
    }
    VBX_PHASE(work, f, 10);
    const double tot = group_sum<4>(up_sum + down_sum);
    // defer (round 6, the fused frame loop): the filter's sum goes into the frame's MFCC row as it is; mfcc_rows_kernel (k_mfcc.hip)
    // takes log10, the clamp and the DCT afterwards, a LANE per row -- ~200 vector instructions of this wavefront (the log's range
    // reduction and polynomial, sixteen table loads, the DCT) for 13 numbers one lane handles.  Same operations in the same order.
    if (defer) {
        if (have && sub == 0) out_row[w] = tot;
        return;
    }
    if (have && sub == 0) {
        const double lg = log10(tot);
        en[w] = (lg != lg || lg < 1.0e-10) ? 1.0e-10 : lg;    // f64::max(1e-10): NaN yields the other operand
    }
    VBX_PHASE(work, f, 11);
    // the lane's DCT row, requested before the log above is needed (num_coeffs <= 16)
    double drow[16];
#pragma unroll
    for (int j = 0; j < 16; j++)                          // (no condition on the loads: clamped indices)
        drow[j] = dct_table[((lane < num_coeffs) ? lane : num_coeffs - 1) * num_coeffs + ((j < num_coeffs) ? j : num_coeffs - 1)];
    wave_sync();
    if (lane < num_coeffs) {                              // dct (:391-397)
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++) if (j < num_coeffs) acc = acc + en[j] * drow[j];
        out_row[lane] = 2.0 * acc;
    }
    VBX_PHASE(work, f, 12);
}

// The deferred tail of MFCC::mfcc (src/spectrum.rs:434-439, :391-397; mfcc_tail_q with `defer`) on ONE row by ONE lane: the row holds its
// num_coeffs (<= 16) mel filter sums; log10 clamped at 1e-10 (f64::max: NaN yields the other operand), then the DCT-II x 2, in place.  The
// operations of mfcc_tail_q's last two steps in their order (bit-identical rows: tools/experiments/bitcompare_libs.py); the results leave
// in one burst of stores (thirteen 8-byte stores spread over the DCT's arithmetic were written back line by line: 600 B per row).
__device__ __forceinline__ void mfcc_row_tail(double *__restrict__ r, int num_coeffs, const double *__restrict__ dct_table) {
    double en[16], outv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double tot = r[(j < num_coeffs) ? j : num_coeffs - 1];
        const double lg = log10(tot);
        en[j] = (lg != lg || lg < 1.0e-10) ? 1.0e-10 : lg;
    }
#pragma unroll
    for (int w = 0; w < 16; w++) {
        double acc = 0.0;
        const int wr = (w < num_coeffs) ? w : num_coeffs - 1;
#pragma unroll
        for (int j = 0; j < 16; j++) if (j < num_coeffs) acc = acc + en[j] * dct_table[wr * num_coeffs + j];
        outv[w] = 2.0 * acc;
    }
#pragma unroll
    for (int w = 0; w < 16; w++) if (w < num_coeffs) r[w] = outv[w];
}

}  // namespace vbx
