// k_burg_fast_p14.hip -- the one-pass Burg kernels at order 14 (vbx_burg_fast.hpp)
#include "vbx_burg_fast.hpp"

namespace vbx {

VBX_BURG_FAST_INSTANTIATE(14)

}  // namespace vbx
