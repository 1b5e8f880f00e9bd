// k_lpc_exact.hip -- LPC::lpc (src/spectrum.rs:63-84) on Autocorrelate::autocorrelate (src/periodic.rs:276-289) of the frames
// whose Levinson row the fused kernels' probe found ill-conditioned (levinson_probe, vbx_spectral.hpp): the p + 1 lag sums
// accumulated in double-double (Ogita / Rump / Oishi's Dot2: every product split exactly by one FMA, every addition's rounding
// recovered by TwoSum), the recursion in double-double, one rounding at the end.  What comes out is the exact-arithmetic row of
// the f64 frame rounded to f64 -- closer to the exact answer than either f64 recursion (the reference's own included) can be on
// such a frame, and what tests/test_gpu_soak.py holds against the same recursion in long double.
//
// Row A10 of SURVEY 8a.  Cost per listed frame: ~10 instructions per product (13 n products over 64 lanes: 2.4 k at 1200
// samples) + a sixteenth of the serial recursion -- a fifth of a frame of analyze_kernel; the list is ~0.1 % of the synthetic
// signal's frames and ~10 % of real 44.1 kHz speech at order 13 (tools/experiments notes in DESIGN.md section 3).
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

namespace {

struct dd { double h, l; };

// error-free transformations; nothing here may be contracted or reassociated
__device__ __forceinline__ dd two_sum(double a, double b) {
#pragma clang fp contract(off)
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd fast_two_sum(double a, double b) {      // |a| >= |b| (or a == 0)
#pragma clang fp contract(off)
    const double s = a + b;
    return dd{s, b - (s - a)};
}
__device__ __forceinline__ dd two_prod(double a, double b) {
    const double p = a * b;
    return dd{p, fma(a, b, -p)};
}
__device__ __forceinline__ dd dd_add(dd x, dd y) {
#pragma clang fp contract(off)
    dd s = two_sum(x.h, y.h);
    const dd t = two_sum(x.l, y.l);
    s.l = s.l + t.h;
    s = fast_two_sum(s.h, s.l);
    s.l = s.l + t.l;
    return fast_two_sum(s.h, s.l);
}
__device__ __forceinline__ dd dd_neg(dd x) { return dd{-x.h, -x.l}; }
__device__ __forceinline__ dd dd_mul(dd x, dd y) {
#pragma clang fp contract(off)
    dd p = two_prod(x.h, y.h);
    p.l = p.l + (x.h * y.l + x.l * y.h);
    return fast_two_sum(p.h, p.l);
}
__device__ __forceinline__ dd dd_div(dd x, dd y) {            // three quotient digits from ONE reciprocal: ~1e-32 relative
#pragma clang fp contract(off)
    const double yi = 1.0 / y.h;
    const double q1 = x.h * yi;
    dd r = dd_add(x, dd_neg(dd_mul(y, dd{q1, 0.0})));
    const double q2 = r.h * yi;
    r = dd_add(r, dd_neg(dd_mul(y, dd{q2, 0.0})));
    const double q3 = r.h * yi;
    dd q = fast_two_sum(q1, q2);
    return dd_add(q, dd{q3, 0.0});
}
// sum of n <= 16 double-doubles as a balanced tree: four dependent additions instead of fifteen (a listed frame is a latency:
// its wavefront is alone on its SIMD, nothing else covers a chain)
template <int N>
__device__ __forceinline__ dd dd_tree_sum(dd (&t)[N], const int n) {      // n <= N entries (n known after unrolling: the guards fold away)
#pragma unroll
    for (int stride = 1; stride < N; stride *= 2) {
#pragma unroll
        for (int i = 0; i + stride < N; i += 2 * stride) if (i + stride < n) t[i] = dd_add(t[i], t[i + stride]);
    }
    return t[0];
}

constexpr int LX_FPW = 16;            // frames per wavefront pass: their recursions run one per lane afterwards
constexpr int LX_KB = 16;             // lags per accumulation pass (the accumulators live in registers)
constexpr int LX_NLMAX = 32;          // lags per frame: orders up to 31 (VBX_LPC_EXACT_MAX_ORDER)

}  // namespace

// smem: xs [n + LX_NLMAX + 1 (+1)] the windowed frame (zero tail) | part [LX_KB][65][2] the lanes' partial sums |
//       rsum [LX_NLMAX][LX_FPW][2] the lag sums of the pass's frames | ac, tmp [LX_NLMAX][LX_FPW][2] the recursion's rows (lane = frame)
__global__ __launch_bounds__(64) void lpc_exact_list_kernel(const int32_t *__restrict__ frame_list, const int32_t *__restrict__ list_count,
                                                            const double *__restrict__ frames, int n, long stride,
                                                            const double *__restrict__ window, int pcm, int nl,
                                                            double *__restrict__ out_lpc, long lpc_ld) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = lane_id();
    const int count = *list_count;
    const int nx = n + LX_NLMAX + 1;
    double *xs = smem;
    double *part = smem + ((nx + 1) & ~1);
    dd *rsum = reinterpret_cast<dd *>(part + LX_KB * 65 * 2);
    dd *acs = rsum + LX_NLMAX * LX_FPW, *tms = acs + LX_NLMAX * LX_FPW;
    const int seg = (n + 63) / 64;                           // samples per lane
    const int p = nl - 1;
    // frames per pass: sixteen when the list is long (their recursions then run sixteen lanes wide), fewer when every workgroup can
    // have its own (a short list is a latency, not a throughput: 1,000 listed frames of a million must not take sixteen frames' time)
    int fpw = (count + (int)gridDim.x - 1) / (int)gridDim.x;
    fpw = fpw < 1 ? 1 : fpw > LX_FPW ? LX_FPW : fpw;
    for (int c0 = blockIdx.x * fpw; c0 < count; c0 += gridDim.x * fpw) {
        const int nf = (count - c0 < fpw) ? count - c0 : fpw;
        for (int q = 0; q < nf; q++) {
            const long f = (long)frame_list[c0 + q];
            // ---- the windowed frame, exactly the f64 values the caller's kernel summed (one product per sample) ----
            const double *xf = frames + f * stride;
            const int16_t *x16 = reinterpret_cast<const int16_t *>(frames) + f * stride;
            wave_sync();
            for (int i = lane; i < nx; i += 64) {
                double v = 0.0;
                if (i < n) {
                    const double xv = pcm ? pcm16_value(x16[i]) : xf[i];
                    v = (window != nullptr) ? xv * window[i] : xv;
                }
                xs[i] = v;
            }
            // ---- r[k] = x[0] + sum_{i = 1}^{n - k - 1} x[i] x[i + k] (Q1: the fold's seed), lane l: i in [1 + l seg, 1 + (l + 1) seg),
            //      LX_KB lags per pass ----
            for (int k0 = 0; k0 < nl; k0 += LX_KB) {
                wave_sync();
                double sh[LX_KB], sl[LX_KB];
#pragma unroll
                for (int k = 0; k < LX_KB; k++) { sh[k] = 0.0; sl[k] = 0.0; }
                const int i0 = 1 + lane * seg;
                for (int j = 0; j < seg; j++) {
                    const int i = i0 + j;
                    if (i >= n) break;                       // (the zero tail makes i + k >= n contribute exactly nothing)
                    const double xi = xs[i];
#pragma unroll
                    for (int k = 0; k < LX_KB; k++) {
#pragma clang fp contract(off)
                        const dd pr = two_prod(xi, xs[i + k0 + k]);
                        const dd s = two_sum(sh[k], pr.h);
                        sh[k] = s.h;
                        sl[k] = sl[k] + (s.l + pr.l);
                    }
                }
#pragma unroll
                for (int k = 0; k < LX_KB; k++) { part[(k * 65 + lane) * 2] = sh[k]; part[(k * 65 + lane) * 2 + 1] = sl[k]; }
                wave_sync();
                {                                            // lane 4 k + q: sixteen of lag k's 64 partial sums, then the four quarters, then the seed
#pragma clang fp contract(off)
                    const int k = lane >> 2, qt = lane & 3;
                    double h = 0.0, l = 0.0;
#pragma unroll 4
                    for (int u = 16 * qt; u < 16 * qt + 16; u++) {
                        const dd s = two_sum(h, part[(k * 65 + u) * 2]);
                        h = s.h;
                        l = l + (s.l + part[(k * 65 + u) * 2 + 1]);
                    }
                    wave_sync();
                    part[(k * 65 + qt) * 2] = h; part[(k * 65 + qt) * 2 + 1] = l;
                    wave_sync();
                    if (qt == 0 && k0 + k < nl) {
                        h = 0.0; l = 0.0;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const dd s = two_sum(h, part[(k * 65 + u) * 2]);
                            h = s.h;
                            l = l + (s.l + part[(k * 65 + u) * 2 + 1]);
                        }
                        const dd s = two_sum(h, xs[0]);
                        rsum[(k0 + k) * LX_FPW + q] = fast_two_sum(s.h, l + s.l);
                    }
                }
            }
        }
        wave_sync();
        // ---- the recursion of src/spectrum.rs:63-84 in double-double, one listed frame per lane ----
        if (lane < nf) {
            double *row = out_lpc + (long)frame_list[c0 + lane] * lpc_ld;
            if (nl == 13) {                                  // the fused kernels' order: rows in registers (a short list is this chain's latency)
                constexpr int P12 = 12;
                dd r[P12 + 1], ac[P12 + 1], tmp[P12 + 1];
#pragma unroll
                for (int k = 0; k <= P12; k++) r[k] = rsum[k * LX_FPW + lane];
                dd err = r[0];
                ac[0] = dd{1.0, 0.0};
#pragma unroll
                for (int k = 1; k <= P12; k++) ac[k] = dd{0.0, 0.0};
#pragma unroll
                for (int i = 1; i <= P12; i++) {
                    dd terms[P12 + 1];                       // r[i] and the i - 1 products, summed as a tree (any order: 1e-30)
                    terms[0] = r[i];
#pragma unroll
                    for (int j = 1; j <= P12; j++) terms[j] = (j < i) ? dd_mul(ac[j], r[i - j]) : dd{0.0, 0.0};
                    const dd acc = dd_tree_sum<P12 + 1>(terms, i);
                    const dd k = dd_div(dd_neg(acc), err);
                    ac[i] = k;
#pragma unroll
                    for (int j = 0; j <= P12; j++) tmp[j] = ac[j];
#pragma unroll
                    for (int j = 1; j < i; j++) ac[j] = dd_add(tmp[j], dd_mul(k, tmp[i - j]));
                    err = dd_mul(err, dd_add(dd{1.0, 0.0}, dd_neg(dd_mul(k, k))));
                }
#pragma unroll
                for (int k = 0; k <= P12; k++) row[k] = ac[k].h + ac[k].l;
            } else {                                         // any order: the rows in LDS, entry k at [k * LX_FPW]
                dd *r = rsum + lane, *ac = acs + lane, *tmp = tms + lane;
                dd err = r[0];
                ac[0] = dd{1.0, 0.0};
                for (int k = 1; k <= p; k++) ac[k * LX_FPW] = dd{0.0, 0.0};
                for (int i = 1; i <= p; i++) {
                    dd acc = r[i * LX_FPW];
                    for (int j = 1; j < i; j++) acc = dd_add(acc, dd_mul(ac[j * LX_FPW], r[(i - j) * LX_FPW]));
                    const dd k = dd_div(dd_neg(acc), err);
                    ac[i * LX_FPW] = k;
                    for (int j = 0; j <= p; j++) tmp[j * LX_FPW] = ac[j * LX_FPW];
                    for (int j = 1; j < i; j++) ac[j * LX_FPW] = dd_add(tmp[j * LX_FPW], dd_mul(k, tmp[(i - j) * LX_FPW]));
                    err = dd_mul(err, dd_add(dd{1.0, 0.0}, dd_neg(dd_mul(k, k))));
                }
                for (int k = 0; k <= p; k++) row[k] = ac[k * LX_FPW].h + ac[k * LX_FPW].l;
            }
        }
        wave_sync();
    }
}

bool lpc_exact_supported(int n, int p) { return p >= 1 && p + 1 <= LX_NLMAX && n >= 2 && n <= 4096; }

void launch_lpc_exact_list(hipStream_t s, const int32_t *frame_list, const int32_t *list_count, int cus, const double *x, int n,
                           long stride, const double *window, bool pcm, int p, double *out_lpc, long lpc_ld) {
    const size_t nx = (size_t)n + LX_NLMAX + 1;
    // order 12 runs its recursion in registers: no rows in LDS (acs / tms), more workgroups per CU
    const size_t rows = (p + 1 == 13) ? 1 : 3;
    const size_t lds = (((nx + 1) & ~(size_t)1) + LX_KB * 65 * 2) * sizeof(double) + rows * (size_t)LX_NLMAX * LX_FPW * sizeof(dd);
    // A grid the chip holds AT ONCE (LDS-limited workgroups per CU): the list's length is only known on the device, every workgroup
    // takes ceil(count / grid) frames (<= 16 per pass), and a frame is ~10-30 us of dependent double-double arithmetic -- with a grid
    // of several residencies the first one's workgroups held all the work and the kernel lasted two or three of them (round 6: config
    // 2's 1,000 listed frames took 70 us on a grid of 2,048).
    size_t per_cu = (160 * 1024) / lds;
    per_cu = per_cu < 1 ? 1 : per_cu > 8 ? 8 : per_cu;
    const unsigned grid = (unsigned)((cus > 0 ? cus : 256) * per_cu);
    hipLaunchKernelGGL(lpc_exact_list_kernel, dim3(grid), dim3(64), lds, s, frame_list, list_count, x, n, stride, window,
                       pcm ? 1 : 0, p + 1, out_lpc, lpc_ld);
}

}  // namespace vbx
