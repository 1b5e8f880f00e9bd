// k_spectral_pow2.hip -- host side of the power-of-two spectral kernels (vbx_spectral_pow2.hpp): twiddle tables and the
// dispatch over the transform size.  The three sizes are instantiated in k_spectral_pow2_u{1,2,4}.hip so that they compile
// in parallel (the 4096-point one alone takes over a minute).
#include "vbx_spectral_pow2.hpp"

namespace vbx {

int spectral_pow2_tab_complex(int plan) {
    return plan == SPECTRAL_PLAN_1024 ? pow2_geom<1>::TAB : plan == SPECTRAL_PLAN_2048 ? pow2_geom<2>::TAB : pow2_geom<4>::TAB;
}

// twiddles, evaluated in long double and rounded once: h_out[2 * spectral_pow2_tab_complex(plan)]
template <int U>
static void fill_tab(double *h) {
    using G = pow2_geom<U>;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    size_t o = 0;
    auto put = [&](long num, long den) {             // e^{-2 pi i num / den}
        const long double ang = two_pi * (long double)(num % den) / (long double)den;
        h[o++] = (double)cosl(ang); h[o++] = (double)(-sinl(ang));
    };
    for (long np = 0; np < G::UNITS; np++) for (long ka = 0; ka < 16; ka++) put(np * ka, G::NC);
    for (long c = 0; c < G::R; c++) for (long kb = 0; kb < 16; kb++) put(c * kb, 16 * G::R);
    for (long m = 0; m <= G::NC / 2; m++) put(m, 2 * G::NC);
}
void spectral_pow2_fill_tab(int plan, double *h_out) {
    if (plan == SPECTRAL_PLAN_1024) fill_tab<1>(h_out); else if (plan == SPECTRAL_PLAN_2048) fill_tab<2>(h_out); else fill_tab<4>(h_out);
}

// The split form's second and third kernels (SP_ANALYZE_SPLIT): what the fused kernel does at the end of its life, from the frame's lag
// curve in its scratch row.
//   scan_curve_kernel    the peak scan and the frequency filter (pitch_refine_store<1>) over the whole stored curve (n / 2 + 2 lags or more)
//                        where it lies (just written: L2); only the candidate list in LDS; the filtered candidates' lags to the frame's
//                        list in HBM.
//   refine_list_kernel   only the lags a candidate's refinement reads (pitch_curve_reach: 2 sr / fmin + 16, 1,296 at speech settings)
//                        and the list into LDS -- 15 KB instead of 26 at 4096 samples: eleven frames per CU instead of six, where the
//                        4096-point transform's exchange buffer leaves the fused kernel four, one refining wavefront per SIMD --,
//                        first-evaluation bounds and refinement (pitch_refine_store<2>).
// A frame whose peak or filter decisions lie within the curve's error goes to the direct-sum fallback list (the fused kernel's own test).
// A frame with a candidate whose PEAK lies beyond the lags refine_list_kernel holds -- the reference's parabolic lag, :423-425 with Q5's
// sign, can land inside the searched band from a peak far outside it when the curve is nearly flat there: 2 % of the bench signal's
// frames -- goes to refine_far_kernel: the whole curve in LDS, every stage in one call, as the fused kernel does it.
__global__ __launch_bounds__(64) void scan_curve_kernel(const spectral_args_t a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];     // the candidate list only: (n / 4 + 8) uint16 -- the curve is read where it lies
    const long fb = xcd_item(blockIdx.x, a.n_batch);
    if (fb >= a.n_batch) return;
    const long f = a.f0 + fb;
    const int lane = lane_id();
    double *row = a.curve + fb * a.curve_ld;
    const double unc_tol = a.curve_tol[fb];
    cand_t *cl = reinterpret_cast<cand_t *>(smem);
    int ncand = 0;
    const bool ok = pitch_refine_store<1>(row, a.n, a.pp, f, a.out_cand, a.cand_ld, a.out_count, a.pitch_status, a.work, unc_tol, nullptr, &ncand, 0, cl);
    const int k_ok = (a.reach - 16) / 2 + 2;                 // peaks up to here have all of their refinement's lags below `reach`
    bool far = false;
    if (ok) for (int i = lane; i < ncand; i += 64) far = far || (int)cl[i] > k_ok;
    int32_t *out = a.curve_list + fb * a.list_ld;
    if (!ok) { if (lane == 0) { out[0] = -1; a.unsure_list[atomicAdd(a.unsure_count, 1)] = (int32_t)f; } return; }
    if (ncand > a.cand_cap || __any(far)) { if (lane == 0) { out[0] = -2; a.far_list[4 + atomicAdd(a.far_list, 1)] = (int32_t)fb; } return; }
    if (lane == 0) out[0] = ncand;
    cand_t *ol = reinterpret_cast<cand_t *>(out + 1);
    for (int i = lane; i < ncand; i += 64) ol[i] = cl[i];
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void refine_list_kernel(const spectral_args_t a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long fb = xcd_item(blockIdx.x, a.n_batch);
    if (fb >= a.n_batch) return;
    const long f = a.f0 + fb;
    const int lane = lane_id();
    const int32_t *in = a.curve_list + fb * a.list_ld;
    int ncand = in[0];
    if (ncand < 0) return;                                   // (the fallback list has it)
    const int nst = a.reach;                                 // even
    const double2 *row = reinterpret_cast<const double2 *>(a.curve + fb * a.curve_ld);
    double2 *ys2 = reinterpret_cast<double2 *>(smem);
    for (int i = lane; i < nst / 2; i += 64) ys2[i] = row[i];
    if (lane < Y_PAD) smem[nst + lane] = 0.0;                // zeros from the cut on, as past the frame
    const int nblk = (nst + PB - 1) / PB;
    cand_t *cl = reinterpret_cast<cand_t *>(reinterpret_cast<float *>(smem + nst + Y_PAD + ((nblk + 2) & ~1)) + a.cand_cap);
    const cand_t *il = reinterpret_cast<const cand_t *>(in + 1);
    for (int i = lane; i < ncand; i += 64) cl[i] = il[i];
    wave_sync();
    pitch_params_t pp = a.pp;
    pp.ncurve = nst;
    pitch_refine_store<2>(smem, a.n, pp, f, a.out_cand, a.cand_ld, a.out_count, a.pitch_status, a.work, 0.0, nullptr, &ncand, a.cand_cap);
}

// the frames of far_list: the whole stored curve, every stage (what the fused kernel does after its transforms); a fixed grid over a count
// only the device knows
__global__ __launch_bounds__(64) void refine_far_kernel(const spectral_args_t a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = lane_id();
    const int count = a.far_list[0];
    const int nst = a.pp.ncurve > 0 ? a.pp.ncurve : a.n;    // (an odd n: the whole curve; its row ends in one more zero)
    for (int i = (int)blockIdx.x; i < count; i += (int)gridDim.x) {
        const long fb = a.far_list[4 + i];
        const long f = a.f0 + fb;
        const double2 *row = reinterpret_cast<const double2 *>(a.curve + fb * a.curve_ld);
        double2 *ys2 = reinterpret_cast<double2 *>(smem);
        wave_sync();                                         // the previous frame's reads
        for (int j = lane; j < (nst + Y_PAD + 1) / 2; j += 64) ys2[j] = row[j];
        const double unc_tol = a.curve_tol[fb];
        wave_sync();
        if (!pitch_refine_store(smem, a.n, a.pp, f, a.out_cand, a.cand_ld, a.out_count, a.pitch_status, a.work, unc_tol, nullptr)) {
            if (lane == 0) a.unsure_list[atomicAdd(a.unsure_count, 1)] = (int32_t)f;
        }
    }
}

void launch_refine_curve(hipStream_t s, const spectral_args_t &a, size_t lds_scan, size_t lds_refine) {
    (void)hipMemsetAsync(a.far_list, 0, sizeof(int32_t), s);
    const size_t lds_list = ((size_t)(a.n / 4 + 8) * sizeof(cand_t) + 15) & ~(size_t)15;
    hipLaunchKernelGGL(scan_curve_kernel, dim3((unsigned)a.n_batch), dim3(64), lds_list, s, a);
    hipLaunchKernelGGL(refine_list_kernel, dim3((unsigned)a.n_batch), dim3(64), lds_refine, s, a);
    const unsigned gf = a.n_batch < 2048 ? (unsigned)a.n_batch : 2048u;
    hipLaunchKernelGGL(refine_far_kernel, dim3(gf), dim3(64), lds_scan, s, a);
}

// bytes per frame of the scratch between the kernels (the cut curve + its zeros, one tolerance, the candidate list), 0 where there is no split form
size_t spectral_split_row_bytes(int n, double sample_rate, double fmin) {
    if (spectral_plan(n) != SPECTRAL_PLAN_4096) return 0;
    int nst = pitch_curve_entries(n, sample_rate, fmin);
    const int reach = pitch_curve_reach(n, sample_rate, fmin);
    if (nst <= 0 && (n & 1)) nst = n + 1;                    // an odd length's whole curve
    if (nst <= 0 || reach <= 0) return 0;
    return (size_t)(nst + Y_PAD + 1) * sizeof(double) + (spectral_split_list_ints(reach) + 1) * sizeof(int32_t) + 16;    // (+ far_list: one index per frame, the count)
}

int launch_pow2_u1(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a);
int launch_pow2_u2(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a);
int launch_pow2_u4(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a);

int launch_analyze_pow2(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a) {
    return L.plan == SPECTRAL_PLAN_1024 ? launch_pow2_u1(s, L, a) : L.plan == SPECTRAL_PLAN_2048 ? launch_pow2_u2(s, L, a) : launch_pow2_u4(s, L, a);
}

}  // namespace vbx
