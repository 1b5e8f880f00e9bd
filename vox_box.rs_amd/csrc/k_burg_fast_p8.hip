// k_burg_fast_p8.hip -- the one-pass Burg kernels at order 8 (vbx_burg_fast.hpp)
#include "vbx_burg_fast.hpp"

namespace vbx {

VBX_BURG_FAST_INSTANTIATE(8)

}  // namespace vbx
