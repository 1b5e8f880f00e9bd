// k_spectral_pow2_u2.hip -- analyze_pow2_kernel<2, ..>: complex FFT of 2048 (vbx_spectral_pow2.hpp)
#include "vbx_spectral_pow2.hpp"

namespace vbx {

#ifdef VBX_POW2_2048_TWO_WAVES
int launch_pow2_u2(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a) { return launch_pow2_u<1, 2>(s, L, a); }
#else
int launch_pow2_u2(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a) { return launch_pow2_u<2>(s, L, a); }
#endif

}  // namespace vbx
