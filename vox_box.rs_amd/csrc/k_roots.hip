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
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

constexpr int ROOTS_BLOCK = 64;


template <typename T>
struct lds_poly_t {
    cx<T> *base;   // element j of this lane at base[j * ROOTS_BLOCK]
    __device__ __forceinline__ cx<T> get(int j) const { return base[j * ROOTS_BLOCK]; }
    __device__ __forceinline__ void set(int j, cx<T> v) const { base[j * ROOTS_BLOCK] = v; }
};
using lds_poly = lds_poly_t<double>;

// src/polynomial.rs:26-32
template <typename T>
__device__ __forceinline__ int poly_degree(const lds_poly_t<T> &p, int len) {
    int d = 0;
    for (int j = 0; j < len; j++) if (!ciszero(p.get(j))) d = j;
    return d;
}
template <typename T>
__device__ __forceinline__ int poly_off_low(const lds_poly_t<T> &p, int len) {
    int d = -1;
    for (int j = len - 1; j >= 0; j--) if (!ciszero(p.get(j))) d = j;
    return d < 0 ? 0 : d;
}

// src/polynomial.rs:34-72.  n = len - 1 stays fixed across deflations (Q11).
// top (<= n): every coefficient above index `top` is zero (deflation clears them).  The reference's Horner chains run
// through those zeros -- 0 * z + 0 three times per index, which leaves all three accumulators exactly zero for finite
// z -- so starting at `top` is the same arithmetic with the no-ops left out (37 % of the Horner steps of an order-12
// polynomial).  A non-finite z (0 * inf = NaN in the reference) takes the full chain.
template <typename T>
__device__ __forceinline__ cx<T> laguerre(const lds_poly_t<T> &p, int len, cx<T> start, int top = -1) {
    const int n = len - 1;
    const T dn = (T)n, dnn1 = (T)(n - 1) * (T)n;
    cx<T> z = start;
    bool done = false;
    if (top < 0 || top > n) top = n;
    // the highest degree among the ACTIVE lanes, as a scalar: entries above a lane's own degree are zeros, and a chain
    // that starts above it runs through 0 * z + 0 (see above).  The callers reach this point diverged (find_roots_emit's
    // early returns and its per-lane trip count), so the maximum is built bit by bit from ballots, which only ever see
    // the active lanes -- a shuffle butterfly would read inactive lanes' registers.
    int top_u = 0;
    for (int bit = 64; bit > 0; bit >>= 1)                   // top <= n <= 64 (len <= 65 by the API's bound)
        if (__any(top >= (top_u | bit))) top_u |= bit;
    for (int it = 0; it < 20; it++) {
        const bool zfin = (z.re - z.re == T(0)) && (z.im - z.im == T(0));
        cx<T> a0, a1 = cmk<T>(T(0), T(0)), a2 = cmk<T>(T(0), T(0));
        if (__all(zfin || done)) {
            // the usual case: the chain starts at the wave's highest degree, a scalar -- the loop runs on scalar control
            // (unrolled by two: the three accumulators rotate through the same registers without copies)
            const int hi = top_u;
            a0 = p.get(hi);
#pragma unroll 2
            for (int j = hi - 1; j >= 0; j--) {
                a2 = cmad(a2, z, a1);
                a1 = cmad(a1, z, a0);
                a0 = cmad(a0, z, p.get(j));
            }
        } else {
            a0 = p.get(n);                                   // a non-finite z somewhere: every lane's full chain
            for (int j = n - 1; j >= 0; j--) {
                a2 = cmad(a2, z, a1);
                a1 = cmad(a1, z, a0);
                a0 = cmad(a0, z, p.get(j));
            }
        }
        // |p(z)| <= 1e-16  (compared on squared norms)
        const T n0 = a0.re * a0.re + a0.im * a0.im;
        if (!done && n0 <= T(1.0e-32)) done = true;
        if (__all(done)) break;
        const cx<T> ca = cdiv(cneg(a1), a0);
        const cx<T> ca2 = cmul(ca, ca);
        const cx<T> cb = csub(ca2, cdiv(cmk<T>(T(2) * a2.re, T(2) * a2.im), a0));
        const cx<T> c1 = csqrt(csub(cmk<T>(dnn1 * cb.re, dnn1 * cb.im), ca2));
        const cx<T> cc1 = cadd(ca, c1), cc2 = csub(ca, c1);
        const T m1 = cc1.re * cc1.re + cc1.im * cc1.im, m2 = cc2.re * cc2.re + cc2.im * cc2.im;
        const cx<T> den = (m1 > m2) ? cc1 : cc2;
        const cx<T> cc = cdiv(cmk<T>(dn, T(0)), den);
        if (!done) z = cadd(z, cc);
    }
    return z;
}

// src/polynomial.rs:92-152 (+ div_polynomial_mut :155-195 inlined as in-place synthetic division).
// co: polynomial in / scratch; emit(index, root) receives the roots in discovery order.
template <typename T, typename Emit>
__device__ __forceinline__ int find_roots_emit(const lds_poly_t<T> &co, int len, Emit emit) {
    const int coeff_high = poly_degree(co, len);
    if (coeff_high < 1) return 2;                       // Err(Polynomial), :95
    const int coeff_low = poly_off_low(co, len);
    if (coeff_low > 0) return 4;                        // coeffs[co] out of bounds, :110-112
    int m = coeff_high;
    const int clen = coeff_high + 1;
    int zi = 0;
    for (int k = m; k >= 3; k--) {                      // (3..m+1).rev()
        // the lanes of a wave hold different polynomials: the chain starts at the highest degree among them
        const int ns = poly_degree(co, clen);          // this lane's degree (the division below starts from it)
        const cx<T> z = laguerre(co, clen, cmk<T>(T(-2), T(-2)), ns);   // laguerre() takes the wave's maximum of it
        emit(zi++, z);
        if (ciszero(z)) return 2;                       // div by zero -> Err, :123,:192
        // divide by (x - z): other = -z; q[i] = c[i+1] - q[i+1]*other
        cx<T> t = co.get(ns);
        for (int i = ns - 1; i >= 0; i--) {
            const cx<T> old = co.get(i);
            co.set(i, t);
            t = cmad(t, z, old);                        // old - t*(-z)
        }
        co.set(ns, cmk<T>(T(0), T(0)));
        m -= 1;
    }
    if (m == 2) {                                       // :131-139
        const cx<T> c0 = co.get(0), c1 = co.get(1), c2 = co.get(2);
        const cx<T> a2 = cadd(c2, c2);
        const cx<T> four_ac = cmul(cmk<T>(T(4) * c2.re, T(4) * c2.im), c0);
        const cx<T> d = csqrt(csub(cmul(c1, c1), four_ac));
        const cx<T> xx = cneg(c1);
        emit(zi, cdiv(cadd(xx, d), a2));
        emit(zi + 1, cdiv(csub(xx, d), a2));
        zi += 2;
    } else if (m == 1) {                                // :141-144
        emit(zi, cdiv(cneg(co.get(0)), co.get(1)));
        zi += 1;
    }
    return 0;
}

// roots into an LDS array (len entries, zero filled past the roots)
template <typename T>
__device__ __forceinline__ int find_roots_lane(const lds_poly_t<T> &co, const lds_poly_t<T> &zr, int len) {
    for (int j = 0; j < len; j++) zr.set(j, cmk<T>(T(0), T(0)));
    return find_roots_emit(co, len, [&](int i, cx<T> z) { zr.set(i, z); });
}

// src/spectrum.rs:166-192
__device__ __forceinline__ bool resonance_from_root(c64 root, double sample_rate, res_t &out) {
    const double freq_mul = sample_rate / (M_PI * 2.0);
    if (!(root.im >= 0.0)) return false;
    double r = hypot(root.re, root.im), theta = atan2(root.im, root.re);
    if (r > 1.0) {      // root.conj().inv() = (re, im) / |root|^2
        const double ns = root.re * root.re + root.im * root.im;
        const double ire = root.re / ns, iim = root.im / ns;
        r = hypot(ire, iim); theta = atan2(iim, ire);
    }
    const double frequency = freq_mul * theta;
    const double bandwidth = -2.0 * freq_mul * log(r);
    if (frequency > 50.0 && frequency < sample_rate * 0.5 - 50.0) {
        out.frequency = frequency; out.bandwidth = bandwidth;
        return true;
    }
    return false;
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

// insertion of one resonance into a frequency-sorted row (stable: equal keys keep arrival order)
__device__ __forceinline__ void res_insert_sorted(res_t *row, int count, res_t v) {
    int j = count;
    while (j > 0 && row[j - 1].frequency > v.frequency) { row[j] = row[j - 1]; j--; }
    row[j] = v;
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
    int count = 0;
    int rst = 0;
    if (st == 0) {
        // every root goes straight through Resonance::from_root (im > 0 only, src/lib.rs:94-104); the
        // reference appends and then sorts [0..=rpos] by frequency (stable, :105-110) -- all stored
        // frequencies are > 50, so that is a sorted insertion.  No root array is kept.
        rst = find_roots_emit(co, len, [&](int, c64 z) {
            res_t v;
            if (z.im > 0.0 && count < VBX_MAX_RESONANCES_K && resonance_from_root(z, sample_rate, v)) {
                if (active) res_insert_sorted(row, count, v);
                count++;
            }
        });
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
