// vbx_mfcc_interp.hpp -- device side of the MFCC bins interpolated inside the fused spectral kernels (mfcc_interp_t, vbx_kernels.hpp;
// host tables: k_spectral.hip).  A thread forms one of the frame's DFT bins from HT pairs of consecutive bins of the transform:
//     bin = sum_t  c[t].x * Z[j0 + 2 t]  +  c[t].y * Z[j0 + 2 t + 1],      Z in LDS, c in a table that every frame reads.
#pragma once

#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

#ifndef VBX_EXP_INTERP_CH12
#define VBX_EXP_INTERP_CH12 4
#endif
#ifndef VBX_EXP_INTERP_CH16
#define VBX_EXP_INTERP_CH16 4
#endif
#ifndef VBX_EXP_INTERP_CH20
#define VBX_EXP_INTERP_CH20 4
#endif

namespace vbx {

// One bin.  cf: the thread's first pair of taps (stride nt double2 between pairs); zp: Z[j0].
// The taps are requested in chunks of CH pairs (the table lives in L2: 48-200 KB per shape, every frame reads all of it); the Z reads
// (LDS) follow in groups of two pairs, each group finished before the next one's reads may start (the accumulators pinned, the loads
// fenced): the powers of exchange 4 wait in 40 registers meanwhile.  Measured (timing builds that skip a part, 720,000 frames of
// 1199 / 1103 samples): the loop 1.4 / 1.2 ms of 21.8, the staging 0.2, the products + tail 1.3 / 2.3 -- against 1.26 ms for MFCC from
// the transform's own bins at 1200.  Chunks of 4, 8 / 10 or ALL 16 / 20 pairs in flight (VBX_EXP_INTERP_CH*): the same time to 0.1 %
// -- three wavefronts per SIMD cover the round trips -- but 12, 18 and 27 spilled registers in the 168-register instance (5.4 KB of
// scratch writes per frame with all in flight): chunks of four.
template <int HT, int CH = HT>
__device__ __forceinline__ void mfcc_interp_bin(const double2 *cf, int nt, const double2 *zp, double &vr, double &vi) {
    static_assert(HT % CH == 0 && CH % 2 == 0, "chunks of whole groups");
    double ar0 = 0.0, ai0 = 0.0, ar1 = 0.0, ai1 = 0.0;
#pragma unroll
    for (int h = 0; h < HT; h += CH) {
        double2 c[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) c[i] = cf[(h + i) * nt];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int g = 0; g < CH; g += 2) {
            double2 z[4];
#pragma unroll
            for (int i = 0; i < 4; i++) z[i] = zp[2 * (h + g) + i];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                ar0 = fma(c[g + i].x, z[2 * i].x, ar0); ai0 = fma(c[g + i].x, z[2 * i].y, ai0);
                ar1 = fma(c[g + i].y, z[2 * i + 1].x, ar1); ai1 = fma(c[g + i].y, z[2 * i + 1].y, ai1);
            }
            asm volatile("" : "+v"(ar0), "+v"(ai0), "+v"(ar1), "+v"(ai1));
            asm volatile("" ::: "memory");
        }
    }
    vr = ar0 + ar1; vi = ai0 + ai1;
}

// ht = taps / 2: 12, 16 or 20 (mfcc_interp_taps)
__device__ __forceinline__ void mfcc_interp_bin(int ht, const double2 *cf, int nt, const double2 *zp, double &vr, double &vi) {
    if (ht == 16) mfcc_interp_bin<16, VBX_EXP_INTERP_CH16>(cf, nt, zp, vr, vi);
    else if (ht == 20) mfcc_interp_bin<20, VBX_EXP_INTERP_CH20>(cf, nt, zp, vr, vi);
    else mfcc_interp_bin<12, VBX_EXP_INTERP_CH12>(cf, nt, zp, vr, vi);
}

// Z[j - jmin] = X_M[j] e^{2 pi i j c / M} for the thread's bin m of the transform (and its mirror: Z[-j] = conj Z[j]).  rt: the rotation of
// bin m -- the thread's own table entry rot[tid] advanced by rot[NT] per slot (a recurrence of at most 17 steps, ~2e-15) instead of one
// table entry per slot: those loads were ten dependent round trips in the middle of the split (measured: 10 k cycles per frame).
// mirror: the slot can hold bins whose mirror is read (only the thread's FIRST slot: jmin > -taps / 2 >= -NT)
__device__ __forceinline__ void mfcc_interp_stage(double2 *zc, const mfcc_interp_t &ip, int m, double pr, double pi, double2 &rt, double2 step,
                                                  const bool mirror) {
    const double zr = fma(pr, rt.x, -(pi * rt.y)), zi = fma(pr, rt.y, pi * rt.x);
    if (m >= ip.jmin && m <= ip.jmax) zc[m - ip.jmin] = double2{zr, zi};
    if (mirror && m >= 1 && -m >= ip.jmin) zc[-m - ip.jmin] = double2{zr, -zi};
    const double nx = fma(rt.x, step.x, -(rt.y * step.y)), ny = fma(rt.x, step.y, rt.y * step.x);
    rt = double2{nx, ny};
}

}  // namespace vbx
