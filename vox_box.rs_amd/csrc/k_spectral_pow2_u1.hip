// k_spectral_pow2_u1.hip -- analyze_pow2_kernel<1, ..>: complex FFT of 1024 (vbx_spectral_pow2.hpp)
#include "vbx_spectral_pow2.hpp"

namespace vbx {

int launch_pow2_u1(hipStream_t s, const spectral_launch_t &L, spectral_args_t &a) { return launch_pow2_u<1>(s, L, a); }

}  // namespace vbx
