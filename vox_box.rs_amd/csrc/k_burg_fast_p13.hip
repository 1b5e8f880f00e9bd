// k_burg_fast_p13.hip -- the one-pass Burg kernels at order 13 (vbx_burg_fast.hpp)
#include "vbx_burg_fast.hpp"

namespace vbx {

VBX_BURG_FAST_INSTANTIATE(13)

}  // namespace vbx
