// k_burg_fast.hip -- dispatch of the one-pass Burg (vbx_burg_fast.hpp) by order; the kernels are instantiated per order in
// k_burg_fast_p<P>.hip.
#include "vbx_burg_fast.hpp"

namespace vbx {

// the orders with an instantiation: 12 (BASELINE), 10 and 13 (the reference's own callers: tests/lib.rs:23,52,
// examples/formant_extraction/src/main.rs:53), 8, 14, 16
static bool bf_order(int p) { return p == 8 || p == 10 || p == 12 || p == 13 || p == 14 || p == 16; }

__global__ void count_accumulate_kernel(const int32_t *count, int32_t *total, int reset) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = (reset ? 0 : *total) + *count;
}
void launch_count_accumulate(hipStream_t s, const int32_t *count, int32_t *total, bool reset) {
    hipLaunchKernelGGL(count_accumulate_kernel, dim3(1), dim3(64), 0, s, count, total, reset ? 1 : 0);
}

bool burg_fast_supported(int n, int p) {
    const char *e = getenv("VBX_BURG_DIRECT");               // 1: the direct recursion for every frame (A/B, tests)
    const bool off = e && atoi(e) != 0;
    return !off && bf_order(p) && n >= 256 && n <= 4096;     // shorter frames: the direct form is as fast (measured at 130); above 1280
                                                               // samples: the segmented lag kernel (burg_lags_seg_kernel)
}

// workspace: the scratch (the chunk rounded up to whole tiles of 64 frames x 3 (p + 1) doubles), then the list: count, pad, indices [F]
long burg_fast_chunk(long F) { return F < BF_CHUNK ? (F > 0 ? F : 1) : BF_CHUNK; }
static long bf_ld(long F) { return (burg_fast_chunk(F) + 63) & ~63L; }
static size_t bf_scratch_doubles(long F, int p) { return (size_t)(3 * (p + 1)) * (size_t)bf_ld(F); }
size_t burg_fast_scratch_bytes(long F, int p) { return bf_scratch_doubles(F, p) * sizeof(double) + ((size_t)F + 2) * sizeof(int32_t); }
int32_t *burg_fast_list(void *ws, long F, int p) { return (int32_t *)((double *)ws + bf_scratch_doubles(F, p)); }

#define VBX_BF_DISPATCH(CALL)                                     \
    switch (p) {                                                  \
        case 8: CALL(8); break;                                   \
        case 10: CALL(10); break;                                 \
        case 12: CALL(12); break;                                 \
        case 13: CALL(13); break;                                 \
        case 14: CALL(14); break;                                 \
        case 16: CALL(16); break;                                 \
        default: break;                                           \
    }

// items [i0, i0 + m) of the (mapped) batch, m <= burg_fast_chunk(F): lag sums and edge samples into the scratch ...
void launch_burg_lags(hipStream_t s, const double *x, long F, int n, long stride, const double *window, int p,
                      frame_map_t map, long i0, long m, void *ws) {
#define VBX_BF_CALL(PP) launch_burg_lags_p<PP, double>(s, x, F, n, stride, window, map, i0, m, (double *)ws)
    VBX_BF_DISPATCH(VBX_BF_CALL)
#undef VBX_BF_CALL
}
void launch_burg_lags_pcm16(hipStream_t s, const int16_t *x, long F, int n, long stride, const double *window, int p,
                            frame_map_t map, long i0, long m, void *ws) {
#define VBX_BF_CALL(PP) launch_burg_lags_p<PP, int16_t>(s, x, F, n, stride, window, map, i0, m, (double *)ws)
    VBX_BF_DISPATCH(VBX_BF_CALL)
#undef VBX_BF_CALL
}
// ... and the recursion on them: coefficient rows and status 0, or the frame's index appended to the list
void launch_burg_recursion(hipStream_t s, long F, int p, frame_map_t map, long i0, long m, double *out, int32_t *status, void *ws) {
    int32_t *list = burg_fast_list(ws, F, p);
#define VBX_BF_CALL(PP) launch_burg_recursion_p<PP>(s, (const double *)ws, F, map, i0, m, out, status, list)
    VBX_BF_DISPATCH(VBX_BF_CALL)
#undef VBX_BF_CALL
}

}  // namespace vbx
