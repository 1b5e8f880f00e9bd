// k_burg_fast_p10.hip -- the one-pass Burg kernels at order 10 (vbx_burg_fast.hpp)
#include "vbx_burg_fast.hpp"

namespace vbx {

VBX_BURG_FAST_INSTANTIATE(10)

}  // namespace vbx
