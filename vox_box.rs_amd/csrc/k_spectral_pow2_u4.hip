// k_spectral_pow2_u4.hip -- analyze_pow2_kernel<4, ..>: complex FFT of 4096 (vbx_spectral_pow2.hpp)
#include "vbx_spectral_pow2.hpp"

namespace vbx {

void launch_pow2_u4(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a) { launch_pow2_u<4>(s, L, a); }

}  // namespace vbx
