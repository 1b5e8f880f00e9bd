// k_burg_fast_p12.hip -- the one-pass Burg kernels at order 12 (vbx_burg_fast.hpp)
#include "vbx_burg_fast.hpp"

namespace vbx {

VBX_BURG_FAST_INSTANTIATE(12)

}  // namespace vbx
