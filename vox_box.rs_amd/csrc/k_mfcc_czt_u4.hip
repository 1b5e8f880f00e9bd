// k_mfcc_czt_u4.hip -- mfcc_czt_kernel<4>: the chirp-z MFCC on the complex 4096-point transform (vbx_mfcc_czt.hpp)
#include "vbx_mfcc_czt.hpp"

namespace vbx {

void launch_mfcc_czt_u4(hipStream_t s, const double *x, long F, int n, int n1, long stride, const double *window, const double *tab,
                        const double *chirp, const double *bhat, const int32_t *bins, const double *slopes, const double *dct,
                        int num_coeffs, int nb, double *out, long out_ld, int32_t *status, double *cw_scratch) {
    launch_mfcc_czt_u<2, 2>(s, x, F, n, n1, stride, window, tab, chirp, bhat, bins, slopes, dct, num_coeffs, nb, out, out_ld, status, cw_scratch);
}

}  // namespace vbx
