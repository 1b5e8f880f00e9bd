// k_burg_fast_p16.hip -- the one-pass Burg kernels at order 16 (vbx_burg_fast.hpp)
#include "vbx_burg_fast.hpp"

namespace vbx {

VBX_BURG_FAST_INSTANTIATE(16)

}  // namespace vbx
