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

void launch_pow2_u1(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a);
void launch_pow2_u2(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a);
void launch_pow2_u4(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a);

void launch_analyze_pow2(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a) {
    if (L.plan == SPECTRAL_PLAN_1024) launch_pow2_u1(s, L, a); else if (L.plan == SPECTRAL_PLAN_2048) launch_pow2_u2(s, L, a); else launch_pow2_u4(s, L, a);
}

}  // namespace vbx
