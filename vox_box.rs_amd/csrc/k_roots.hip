// k_roots.hip -- Laguerre root finding with deflation, resonances from roots.
//
// Reference: src/polynomial.rs:26-195 (degree, off_low, laguerre, find_roots_mut,
//            div_polynomial_mut; Q11), src/spectrum.rs:165-210 (Resonance::from_root,
//            to_resonance), src/lib.rs:80-110 (find_formants: polynomial build, im > 0 filter, sort).
//
// The per-polynomial work is a short, strictly sequential recurrence (20 fixed Laguerre
// iterations x 3 Horner chains, then synthetic division), so the mapping is ONE LANE PER
// POLYNOMIAL: 64 frames per wavefront, every lane busy, control flow almost uniform.
// Per-lane polynomial/roots arrays live in LDS as [index][lane] (16-byte elements, lane
// fastest) so dynamic indexing costs one conflict-free ds_read_b128, never scratch memory.
#include "vbx_roots.hpp"

namespace vbx {

// roots into an LDS array (len entries, zero filled past the roots)
template <typename T>
__device__ __forceinline__ int find_roots_lane(const lds_poly_t<T> &co, const lds_poly_t<T> &zr, int len) {
    for (int j = 0; j < len; j++) zr.set(j, cmk<T>(T(0), T(0)));
    return find_roots_emit(co, len, [&](int i, cx<T> z) { zr.set(i, z); });
}

// ---- kernels ---------------------------------------------------------------------------------

// T = double: Complex<f64>; T = float: the f32 instantiation.  CT is the matching {re, im} pair of the ABI.
template <typename T, typename CT>
__global__ __launch_bounds__(ROOTS_BLOCK) void find_roots_kernel(CT *__restrict__ polys, long n_polys, int len,
                                                                 int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cx<T> *lds = reinterpret_cast<cx<T> *>(lds_raw);
    const long f = (long)blockIdx.x * ROOTS_BLOCK + threadIdx.x;
    const bool active = f < n_polys;
    lds_poly_t<T> co{lds + threadIdx.x}, zr{lds + (size_t)len * ROOTS_BLOCK + threadIdx.x};
    const long fr = active ? f : n_polys - 1;           // idle lanes shadow the last polynomial
    CT *pp = polys + fr * (long)len;
    for (int j = 0; j < len; j++) co.set(j, cmk<T>(pp[j].re, pp[j].im));
    const int st = find_roots_lane(co, zr, len);
    if (active) {
        if (st == 0) {
            for (int j = 0; j < len; j++) { const cx<T> v = zr.get(j); pp[j].re = v.re; pp[j].im = v.im; }
        }
        if (status != nullptr) status[f] = st;
    }
}

template <typename T, typename CT>
__global__ __launch_bounds__(ROOTS_BLOCK) void laguerre_kernel(const CT *__restrict__ polys, long n_polys, int len,
                                                               CT start, CT *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cx<T> *lds = reinterpret_cast<cx<T> *>(lds_raw);
    const long f = (long)blockIdx.x * ROOTS_BLOCK + threadIdx.x;
    const bool active = f < n_polys;
    lds_poly_t<T> co{lds + threadIdx.x};
    const long fr = active ? f : n_polys - 1;
    const CT *pp = polys + fr * (long)len;
    for (int j = 0; j < len; j++) co.set(j, cmk<T>(pp[j].re, pp[j].im));
    const cx<T> z = laguerre(co, len, cmk<T>(start.re, start.im));
    if (active) { out[f].re = z.re; out[f].im = z.im; }
}

// Polynomial::div_polynomial_mut (src/polynomial.rs:155-195): self / (x + other), quotient left in self, the
// remainder bookkeeping of the reference reproduced literally (degree() re-evaluated after every zeroing).
// One lane per polynomial, straight on global memory (a helper of find_roots in the reference; not a hot path).
__global__ void div_polynomial_kernel(cplx_t *__restrict__ polys, const cplx_t *__restrict__ others, long n_polys, int len,
                                      cplx_t *__restrict__ rem, int32_t *__restrict__ status) {
    const long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_polys) return;
    cplx_t *p = polys + f * (long)len, *r = rem + f * (long)len;
    const cplx_t o = others[f];
    auto degree = [&](const cplx_t *q) { int d = 0; for (int j = 0; j < len; j++) if (!(q[j].re == 0.0 && q[j].im == 0.0)) d = j; return d; };
    for (int i = 0; i < len; i++) r[i] = p[i];
    int st = 0;
    if (o.re == 0.0 && o.im == 0.0) st = 2;                                        // Err(Polynomial), :192
    else {
        const int ns = degree(p);
        for (int i = ns - 1; i >= 0; i--) {                                           // (0..(ns - ds + 1)).rev(), ds = 1
            p[i] = r[1 + i];
            cplx_t t;                                                                 // rem[i] - self[i] * other
            t.re = r[i].re - (p[i].re * o.re - p[i].im * o.im);
            t.im = r[i].im - (p[i].re * o.im + p[i].im * o.re);
            r[i] = t;
        }
        for (int k = 1; k < ns + 1; k++) { const int d = degree(r); r[d].re = 0.0; r[d].im = 0.0; }   // :174-176
        const int l = degree(p);
        const int cnt = (l + 1) - ns;                                                 // (l + 1) - ns - ds + 1
        if (cnt < 0) st = 4;                                                          // usize underflow panics
        for (int k = 0; k < cnt; k++) { const int d = degree(p); p[d].re = 0.0; p[d].im = 0.0; }
    }
    if (status != nullptr) status[f] = st;
}

void launch_div_polynomial(hipStream_t s, cplx_t *polys, const cplx_t *others, long F, int len, cplx_t *rem, int32_t *status) {
    hipLaunchKernelGGL(div_polynomial_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, s, polys, others, F, len, rem, status);
}

// to_resonance (strict_im = 0: roots with im >= 0, src/spectrum.rs:204-209) or the find_formants
// variant (strict_im = 1: im > 0 first, src/lib.rs:94-104).  One lane per row.
__global__ void to_resonance_kernel(const cplx_t *__restrict__ roots, long n_rows, int n_roots, double sample_rate,
                                    int strict_im, res_t *__restrict__ out, int out_stride,
                                    int32_t *__restrict__ out_count, const int32_t *__restrict__ status) {
    const long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_rows) return;
    res_t *row = out + f * (long)out_stride;
    int count = 0;
    const bool ok = (status == nullptr) || status[f] == 0;
    if (ok) {
        const cplx_t *rr = roots + f * (long)n_roots;
        for (int j = 0; j < n_roots && count < out_stride; j++) {
            const c64 z = cmk(rr[j].re, rr[j].im);
            if (strict_im && !(z.im > 0.0)) continue;
            res_t v;
            if (resonance_from_root(z, sample_rate, v)) { res_insert_sorted(row, count, v); count++; }
        }
    }
    for (int j = count; j < out_stride; j++) { row[j].frequency = 0.0; row[j].bandwidth = 0.0; }
    if (out_count != nullptr) out_count[f] = count;
}

// find_formants core (src/lib.rs:80-110): Burg coefficients -> rev([1, a1..ap]) -> roots ->
// resonances [32] sorted, zero padded.  One lane per frame.
__global__ __launch_bounds__(ROOTS_BLOCK) void formant_resonances_kernel(
    const double *__restrict__ coeffs, long n_frames, int p, double sample_rate,
    res_t *__restrict__ out_res, int32_t *__restrict__ out_count, int32_t *__restrict__ status, const frame_map_t map) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    c64 *lds = reinterpret_cast<c64 *>(lds_raw);
    const long f = frame_map(map, (long)blockIdx.x * ROOTS_BLOCK + threadIdx.x, n_frames);
    const bool active = f >= 0;
    const int len = p + 1;
    lds_poly co{lds + threadIdx.x};
    const long fr = active ? f : n_frames - 1;
    const double *a = coeffs + fr * (long)p;
    int st = (status != nullptr) ? status[fr] : 0;
    // complex_lpc = rev([1, a1..ap]): index j < p holds a[p-1-j], index p holds 1 (src/lib.rs:80-91)
    for (int j = 0; j < p; j++) co.set(j, cmk(a[p - 1 - j], 0.0));
    co.set(p, cmk(1.0, 0.0));
    res_t *row = out_res + fr * (long)VBX_MAX_RESONANCES_K;    // idle lanes shadow the last frame (same values)
    int count = 0, total = 0;
    int rst = 0;
    if (st == 0) {
        // every root goes straight through Resonance::from_root (im > 0 only, src/lib.rs:94-104); the
        // reference appends and then sorts [0..=rpos] by frequency (stable, :105-110) -- all stored
        // frequencies are > 50, so that is a sorted insertion.  No root array is kept.
        rst = find_roots_emit(co, len, [&](int, c64 z) {
            res_t v;
            if (z.im > 0.0 && resonance_from_root(z, sample_rate, v)) {
                if (count < VBX_MAX_RESONANCES_K) { if (active) res_insert_sorted(row, count, v); count++; }
                total++;
            }
        });
        // the 33rd resonance: resonances[count] out of bounds in the reference (src/lib.rs:97-99; reachable from order 34 on,
        // when roots that are real in exact arithmetic come out with tiny positive imaginary parts)
        if (rst == 0 && total > VBX_MAX_RESONANCES_K) rst = 4;   // VBX_FRAME_ERR_PANIC
    }
    if (!active) return;
    if (rst != 0) count = 0;
    for (int j = count; j < VBX_MAX_RESONANCES_K; j++) { row[j].frequency = 0.0; row[j].bandwidth = 0.0; }
    if (out_count != nullptr) out_count[f] = count;
    if (status != nullptr && st == 0 && rst != 0) status[f] = rst;
}

// ---- launchers -----------------------------------------------------------------------------

void launch_find_roots(hipStream_t s, cplx_t *polys, long F, int len, int32_t *status) {
    const size_t lds = (size_t)2 * len * ROOTS_BLOCK * sizeof(c64);
    hipLaunchKernelGGL((find_roots_kernel<double, cplx_t>), dim3((unsigned)((F + ROOTS_BLOCK - 1) / ROOTS_BLOCK)), dim3(ROOTS_BLOCK), lds, s,
                       polys, F, len, status);
}

void launch_laguerre(hipStream_t s, const cplx_t *polys, long F, int len, cplx_t start, cplx_t *out) {
    const size_t lds = (size_t)len * ROOTS_BLOCK * sizeof(c64);
    hipLaunchKernelGGL((laguerre_kernel<double, cplx_t>), dim3((unsigned)((F + ROOTS_BLOCK - 1) / ROOTS_BLOCK)), dim3(ROOTS_BLOCK), lds, s,
                       polys, F, len, start, out);
}

void launch_find_roots_f32(hipStream_t s, cplx32_t *polys, long F, int len, int32_t *status) {
    const size_t lds = (size_t)2 * len * ROOTS_BLOCK * sizeof(c32);
    hipLaunchKernelGGL((find_roots_kernel<float, cplx32_t>), dim3((unsigned)((F + ROOTS_BLOCK - 1) / ROOTS_BLOCK)), dim3(ROOTS_BLOCK), lds, s,
                       polys, F, len, status);
}

void launch_laguerre_f32(hipStream_t s, const cplx32_t *polys, long F, int len, cplx32_t start, cplx32_t *out) {
    const size_t lds = (size_t)len * ROOTS_BLOCK * sizeof(c32);
    hipLaunchKernelGGL((laguerre_kernel<float, cplx32_t>), dim3((unsigned)((F + ROOTS_BLOCK - 1) / ROOTS_BLOCK)), dim3(ROOTS_BLOCK), lds, s,
                       polys, F, len, start, out);
}

void launch_to_resonance(hipStream_t s, const cplx_t *roots, long F, int n_roots, double sample_rate,
                         int strict_im, res_t *out, int out_stride, int32_t *out_count, const int32_t *status) {
    const int bs = 64;
    hipLaunchKernelGGL(to_resonance_kernel, dim3((unsigned)((F + bs - 1) / bs)), dim3(bs), 0, s,
                       roots, F, n_roots, sample_rate, strict_im, out, out_stride, out_count, status);
}

void launch_formant_resonances(hipStream_t s, const double *coeffs, long F, int p, double sample_rate,
                               res_t *out_res, int32_t *out_count, int32_t *status, frame_map_t map) {
    const size_t lds = (size_t)(p + 1) * ROOTS_BLOCK * sizeof(c64);
    const long items = frame_map_items(map, F);
    hipLaunchKernelGGL(formant_resonances_kernel, dim3((unsigned)((items + ROOTS_BLOCK - 1) / ROOTS_BLOCK)), dim3(ROOTS_BLOCK), lds, s,
                       coeffs, F, p, sample_rate, out_res, out_count, status, map);
}

}  // namespace vbx
