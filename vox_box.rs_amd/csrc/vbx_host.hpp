// vbx_host.hpp -- the library's host-only arithmetic (vbx_host.cpp): tables, bins and shard geometry that need no GPU.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace vbx {

// sample 0.10 window / HanningLag / periodic Hanning tables by the crate's own recurrences (VBX_WINDOW_* kinds)
int window_table_host(int kind, size_t n, double *out);
// mel filter bank bins, src/spectrum.rs:411-414 (Q14); overflow: a bin beyond any spectrum (the reference panics)
void mel_bins_host(size_t n, size_t k, double lo, double hi, double sr, std::vector<int32_t> &bins, bool &overflow);

}  // namespace vbx
