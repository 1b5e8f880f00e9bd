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

// The second kernel of the split form (SP_ANALYZE_SPLIT): one wavefront per frame takes the frame's lag curve from its scratch row
// into LDS and runs the refinement (pitch_refine_store) -- what the fused kernel does at the end of its life, at twelve frames per
// CU instead of the four the 4096-point transform's exchange buffer leaves room for.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void refine_curve_kernel(const spectral_args_t a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long fb = xcd_item(blockIdx.x, a.n_batch);
    if (fb >= a.n_batch) return;
    const long f = a.f0 + fb;
    const int lane = lane_id();
    const int nst = a.pp.ncurve;                             // even, > 0 (launch_pow2_u)
    const double2 *row = reinterpret_cast<const double2 *>(a.curve + fb * a.curve_ld);
    double2 *ys2 = reinterpret_cast<double2 *>(smem);
    for (int i = lane; i < (nst + Y_PAD) / 2; i += 64) ys2[i] = row[i];
    const double unc_tol = a.curve_tol[fb];
    wave_sync();
    if (!pitch_refine_store(smem, a.n, a.pp, f, a.out_cand, a.cand_ld, a.out_count, a.pitch_status, a.work, unc_tol, nullptr)) {
        if (lane == 0) a.unsure_list[atomicAdd(a.unsure_count, 1)] = (int32_t)f;
    }
}

void launch_refine_curve(hipStream_t s, const spectral_args_t &a, size_t lds) {
    hipLaunchKernelGGL(refine_curve_kernel, dim3((unsigned)a.n_batch), dim3(64), lds, s, a);
}

// bytes per frame of the scratch between the two kernels (the cut curve + its zeros + one tolerance), 0 where there is no split form
size_t spectral_split_row_bytes(int n, double sample_rate, double fmin) {
    if (spectral_plan(n) != SPECTRAL_PLAN_4096) return 0;
    const int nst = pitch_curve_entries(n, sample_rate, fmin);
    return nst > 0 ? (size_t)(nst + Y_PAD + 1) * sizeof(double) : 0;
}

int launch_pow2_u1(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a);
int launch_pow2_u2(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a);
int launch_pow2_u4(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a);

int launch_analyze_pow2(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a) {
    return L.plan == SPECTRAL_PLAN_1024 ? launch_pow2_u1(s, L, a) : L.plan == SPECTRAL_PLAN_2048 ? launch_pow2_u2(s, L, a) : launch_pow2_u4(s, L, a);
}

}  // namespace vbx
