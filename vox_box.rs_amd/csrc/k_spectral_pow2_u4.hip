// k_spectral_pow2_u4.hip -- complex FFT of 4096: analyze_pow2_kernel<2, .., W = 2>, two wavefronts per frame (vbx_spectral_pow2.hpp)
#include "vbx_spectral_pow2.hpp"

namespace vbx {

int launch_pow2_u4(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a) { return launch_pow2_u<2, 2>(s, L, a); }

}  // namespace vbx
