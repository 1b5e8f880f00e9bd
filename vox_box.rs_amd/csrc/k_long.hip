// k_long.hip -- frames LONGER than the register / LDS-resident kernels hold (more than VBX_MAX_FRAME_LEN_K samples).
//
// The reference's traits take slices of any length (src/periodic.rs:276, src/spectrum.rs:101, src/lib.rs:40), and its own
// integration test hands find_formants a whole 31,232-sample file as ONE frame (tests/lib.rs:14-42).  Such frames do not
// fit a wavefront's registers or one CU's LDS, so these kernels walk them in tiles out of HBM / L2:
//   burg_long_kernel          LPC::lpc_praat_mut (src/spectrum.rs:101-146): one workgroup of 1024 lanes per frame, the two
//                             error arrays ping-pong through an L2-resident scratch, ONE sweep per order (the update of
//                             order i and the sums of order i+1 fused), fixed-order workgroup reductions
//   autocorr_long_kernel      Autocorrelate::autocorrelate_mut (src/periodic.rs:276-289), any number of lags: the FP64
//                             matrix-core lag tiles of vbx_autocorr.hpp over CHUNKED LDS images of the frame (one image for
//                             the A operand, one for the B operand, 1024 samples of the sum at a time), the sum range split
//                             over workgroups when there are few frames; autocorr_long_finish_kernel adds the partial sums
//                             in a fixed order and applies the fold seed (Q1)
//   preemphasis_long_kernel   Filter::preemphasis (src/waves.rs:82-96) of a whole signal: per-tile backward recurrences, a
//                             sequential carry pass over the tile ends, a fix-up pass
//   pitch_long_*_kernel       Pitched::pitch (src/periodic.rs:377-456) on the lag curve in HBM: every lag by autocorr_long, the
//                             curve (normalised, divided by the lag window, zero padded to 2n) as an array, the peak scan with
//                             an order-preserving compaction, improve_extremum per candidate with 16 lanes each (the shared
//                             refinement of vbx_pitch_refine.hpp reading y from memory), a rank sort = the reference's stable sort
//   mfcc_long_*_kernel        MFCC::mfcc (src/spectrum.rs:401-441): the needed bins by the Goertzel-Reinsch recurrence of
//                             k_mfcc.hip, 256 bins per wavefront, the filter sums' inputs in a scratch instead of LDS
// Throughput is not the point of this file (a frame this long is a whole recording, not one of millions), exactness of the
// reference's semantics at every length is; still nothing here is serial over the samples.
#include "vbx_autocorr.hpp"
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"
#include "vbx_mfcc_tail.hpp"
#include "vbx_pitch_refine.hpp"

namespace vbx {

// ---- workgroup sum, bit-identical in every lane: DPP wave sums, then the wave partials added in wave order ------------
template <int BS>
__device__ __forceinline__ double block_sum(double v, double *red /* BS/64 doubles of LDS */) {
    constexpr int NW = BS / 64;
    const double w = wave_sum(v);
    __syncthreads();                                  // the previous use of `red` is over
    if (lane_id() == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NW; k++) s += red[k];
    return s;
}

// ---- Burg on a long frame ------------------------------------------------------------------------------------------------
// ws: per frame of the launch 4 arrays of n doubles (b1 / b2, twice: the update reads one pair and writes the other).
// Order i sums over j in [0, n - i) (src/spectrum.rs:118-121); the update for the next order writes j in [0, n - i - 1)
// from the OLD values at j and j + 1 (:131-134: b1[j] is read before the loop reaches it), which is what makes one sweep per
// order enough: new element -> its products for the next order's sums in the same registers.
template <int BS>
__global__ __launch_bounds__(BS) void burg_long_kernel(const double *__restrict__ x, long f0, long n_frames, long n, long stride,
                                                       const double *__restrict__ window, int p, double *__restrict__ out,
                                                       int32_t *__restrict__ status, double *__restrict__ ws) {
    __shared__ double red[2 * (BS / 64)];
    __shared__ double aa[VBX_MAX_LPC_ORDER_K], co[VBX_MAX_LPC_ORDER_K];
    const long f = f0 + blockIdx.x;
    if (f >= n_frames) return;
    const int t = threadIdx.x;
    const double *xf = x + f * stride;
    double *w0 = ws + (long)blockIdx.x * 4 * n;
    double *ob1 = w0, *ob2 = w0 + n, *nb1 = w0 + 2 * n, *nb2 = w0 + 3 * n;
    double num = 0.0, den = 0.0;
    // b1[j] = x[j], b2[j] = x[j + 1] for j <= n - 2 (:108-114)
    for (long j = t; j < n - 1; j += BS) {
        double v = xf[j], vn = xf[j + 1];
        if (window != nullptr) { v *= window[j]; vn *= window[j + 1]; }
        ob1[j] = v; ob2[j] = vn;
        num = fma(v, vn, num);
        den = fma(v, v, fma(vn, vn, den));
    }
    int st = 0;
    for (int i = 1; i <= p; i++) {
        const double s_num = block_sum<BS>(num, red), s_den = block_sum<BS>(den, red + BS / 64);
        if (s_den <= 0.0) { st = 1; break; }                       // Err(LPC), :123-125 (NaN falls through, as in the reference)
        const double c = 2.0 * s_num / s_den;
        // coeffs[i-1] = c; coeffs[j-1] = aa[j-1] - c * aa[i-j-1], j = 1..i-1 (:126-129)
        if (t < i - 1) co[t] = aa[t] - c * aa[i - 2 - t];
        else if (t == i - 1) co[t] = c;
        __syncthreads();
        if (i == p) break;
        if (t < i) aa[t] = co[t];                                   // :131-133
        num = 0.0; den = 0.0;
        const long m = n - i - 1;                                   // elements of the next order
        for (long j = t; j < m; j += BS) {
            const double t1 = fma(-c, ob2[j], ob1[j]);
            const double t2 = fma(-c, ob1[j + 1], ob2[j + 1]);
            nb1[j] = t1; nb2[j] = t2;
            num = fma(t1, t2, num);
            den = fma(t1, t1, fma(t2, t2, den));
        }
        __syncthreads();                                            // aa is complete; the new pair is written (same workgroup reads it next)
        double *s1 = ob1, *s2 = ob2; ob1 = nb1; ob2 = nb2; nb1 = s1; nb2 = s2;
    }
    __syncthreads();
    if (t < p) out[f * (long)p + t] = (st == 0) ? co[t] * -1.0 : 0.0;   // :142-144
    if (status != nullptr && t == 0) status[f] = st;
}

size_t burg_long_scratch_bytes(long frames, long n) { return (size_t)frames * 4 * (size_t)n * sizeof(double); }

void launch_burg_long(hipStream_t s, const double *x, long f0, long f1, long F, long n, long stride, const double *window,
                      int p, double *out, int32_t *status, double *ws) {
    if (f1 <= f0) return;
    hipLaunchKernelGGL(burg_long_kernel<1024>, dim3((unsigned)(f1 - f0)), dim3(1024), 0, s, x, f0, f1 < F ? f1 : F, n, stride,
                       window, p, out, status, ws);
}

// ---- autocorrelation of a long frame -----------------------------------------------------------------------------------------
// Workgroup (frame f, lag group g, split s): lags [1280 g, 1280 g + 1280) as five matrix-core tiles (vbx_autocorr.hpp), the
// sum index a over the chunks c = s, s + splits, ... of ACL_CHUNK samples.  Two LDS images per chunk, both in the padded layout
// of vbx_autocorr.hpp relative to their own origin (a chunk starts at a multiple of 16, so the pad doubles fall where the
// whole-frame image has them): A holds z[a0 - 256, a0 + ACL_CHUNK + 32), B holds z[a0 + l0, a0 + l0 + ACL_CHUNK + 1280 + 32).
constexpr int ACL_CHUNK = 1024;
constexpr int ACL_A_LOG = ACL_CHUNK + AC_MF_FRONT + 32;                        // logical entries of image A
constexpr int ACL_B_LOG = ACL_CHUNK + AC_MF_NT * AC_MF_TILE + 32;
constexpr int ACL_A_PHYS = ACL_A_LOG + (ACL_A_LOG >> 4) + 1;
constexpr int ACL_B_PHYS = ACL_B_LOG + (ACL_B_LOG >> 4) + 1;
constexpr int ACL_LAGS = AC_MF_NT * AC_MF_TILE;                                // lags per workgroup

template <int L>
__device__ __forceinline__ void acl_chunk(const double *za, const double *zb, int a_count, vbx_d4 (&acc)[AC_MF_NT]) {
    const int lane = lane_id();
    const int row = lane & 15, k = lane >> 4;
    const double *pa[4], *pb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        pa[u] = za + ac_mf_phys(4 * u + k - 16 * row);                          // A: z[a + k - 16 row]
        const int jb = 4 * u + k + row;                                         // B: z[a + k + col + l0], image origin a0 + l0
        pb[u] = zb + jb + (jb >> 4);
    }
    ac_mf_segment<L>(pa, pb, 0, a_count, acc);
}

__global__ __launch_bounds__(64) void autocorr_long_kernel(const double *__restrict__ x, long n_frames, long n, long stride,
                                                           const double *__restrict__ window, long n_lags, int groups,
                                                           int splits, double *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) double za[ACL_A_PHYS], zb[ACL_B_PHYS];
    const long b = blockIdx.x;
    const int g = (int)(b % groups), s = (int)((b / groups) % splits);
    const long f = b / ((long)groups * splits);
    if (f >= n_frames) return;
    const int lane = lane_id();
    const double *xf = x + f * stride;
    const long l0 = (long)g * ACL_LAGS;
    const long want_raw = (n_lags - l0 + AC_MF_TILE - 1) / AC_MF_TILE;
    const int want = (int)(want_raw < AC_MF_NT ? want_raw : AC_MF_NT);           // tiles that hold requested lags
    vbx_d4 acc[AC_MF_NT];
#pragma unroll
    for (int t = 0; t < AC_MF_NT; t++) acc[t] = vbx_d4{0.0, 0.0, 0.0, 0.0};
    const long a_end = n - l0;                                                  // products exist for a < a_end
    for (long a0 = (long)s * ACL_CHUNK; a0 < a_end; a0 += (long)splits * ACL_CHUNK) {
        wave_sync();
        for (int j = lane; j < ACL_A_LOG; j += 64) {
            const long i = a0 - AC_MF_FRONT + j;
            double v = 0.0;
            if (i >= 0 && i < n) { v = xf[i]; if (window != nullptr) v *= window[i]; }
            za[j + (j >> 4)] = v;
        }
        for (int j = lane; j < ACL_B_LOG; j += 64) {
            const long i = a0 + l0 + j;
            double v = 0.0;
            if (i < n) { v = xf[i]; if (window != nullptr) v *= window[i]; }
            zb[j + (j >> 4)] = v;
        }
        wave_sync();
        const long left = a_end - a0;
        const int a_count = (int)(left < ACL_CHUNK ? ((left + 15) & ~15L) : ACL_CHUNK);   // whole 16-sample steps (zeros past n)
        if (want == 5) acl_chunk<5>(za, zb, a_count, acc);
        else if (want == 4) acl_chunk<4>(za, zb, a_count, acc);
        else if (want == 3) acl_chunk<3>(za, zb, a_count, acc);
        else if (want == 2) acl_chunk<2>(za, zb, a_count, acc);
        else acl_chunk<1>(za, zb, a_count, acc);
    }
    double *po = partial + (f * splits + s) * n_lags;
#pragma unroll
    for (int t = 0; t < AC_MF_NT; t++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long lag = l0 + t * AC_MF_TILE + 64 * r + lane;
            if (lag < n_lags) po[lag] = acc[t][r];
        }
    }
}

// r[lag] = (S[lag] - x0 x[lag]) + x0: the fold is seeded with x[0], not x[0] x[lag] (src/periodic.rs:280-287, Q1)
__global__ void autocorr_long_finish_kernel(const double *__restrict__ x, long n_frames, long n, long stride,
                                            const double *__restrict__ window, long n_lags, int splits,
                                            const double *__restrict__ partial, double *__restrict__ out) {
    const long total = n_frames * n_lags;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long f = e / n_lags, lag = e - f * n_lags;
        const double *xf = x + f * stride;
        double x0 = xf[0], xl = xf[lag];
        if (window != nullptr) { x0 *= window[0]; xl *= window[lag]; }
        double sum = 0.0;
        for (int s = 0; s < splits; s++) sum += partial[(f * splits + s) * n_lags + lag];
        out[e] = (sum - x0 * xl) + x0;
    }
}

int autocorr_long_splits(long F, long n, long n_lags) {
    const long groups = (n_lags + ACL_LAGS - 1) / ACL_LAGS, chunks = (n + ACL_CHUNK - 1) / ACL_CHUNK;
    long want = (1280 + F * groups - 1) / (F * groups);                          // five wavefronts per CU
    if (want > chunks) want = chunks;
    if (want > 64) want = 64;
    return (int)(want < 1 ? 1 : want);
}
size_t autocorr_long_scratch_bytes(long F, long n, long n_lags) {
    return (size_t)F * (size_t)autocorr_long_splits(F, n, n_lags) * (size_t)n_lags * sizeof(double);
}
void launch_autocorr_long(hipStream_t s, const double *x, long F, long n, long stride, const double *window, long n_lags,
                          double *out, double *ws) {
    const int groups = (int)((n_lags + ACL_LAGS - 1) / ACL_LAGS), splits = autocorr_long_splits(F, n, n_lags);
    hipLaunchKernelGGL(autocorr_long_kernel, dim3((unsigned)(F * groups * splits)), dim3(64), 0, s, x, F, n, stride, window,
                       n_lags, groups, splits, ws);
    long blocks = (F * n_lags + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(autocorr_long_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, F, n, stride, window, n_lags,
                       splits, (const double *)ws, out);
}

// ---- pre-emphasis of a long signal -----------------------------------------------------------------------------------------
// y[i] = x[i] + c y[i+1] backwards from y[n-1] = x[n-1] (src/waves.rs:88-94).  Tiles of PEL_TILE samples: (1) each tile's
// recurrence with carry-in 0, one lane per run of PEL_RUN samples then a lane scan, exactly the short-frame kernel's scheme
// per tile (k_front.hip); (2) one lane per frame walks the tile ends: carry[T] = y_loc_first(T+1) + c^PEL_TILE carry[T+1];
// (3) y[i] += c^(tile_end - i) carry.  |c| >= 1 (an unstable filter): the reference's own sequential order, one lane per
// frame (the powers of the scan would overflow long before the reference's values do).
constexpr int PEL_TILE = 4096;

__global__ __launch_bounds__(256) void preemphasis_long_local_kernel(const double *__restrict__ x, long n_frames, long n, long stride,
                                                                     double c, double *__restrict__ out, double *__restrict__ heads) {
    __shared__ double sm[PEL_TILE];
    __shared__ double lane_y[256];
    const long tiles = (n + PEL_TILE - 1) / PEL_TILE;
    const long f = blockIdx.x / tiles, T = blockIdx.x % tiles;
    if (f >= n_frames) return;
    const int t = threadIdx.x;
    const long base = T * PEL_TILE;
    const int len = (int)((n - base < PEL_TILE) ? n - base : PEL_TILE);
    const double *xf = x + f * stride + base;
    for (int i = t; i < PEL_TILE; i += 256) sm[i] = (i < len) ? xf[i] : 0.0;
    __syncthreads();
    constexpr int E = PEL_TILE / 256;
    double *mine = sm + t * E;
    double carry = 0.0;
    for (int e = E - 1; e >= 0; e--) { carry = fma(c, carry, mine[e]); mine[e] = carry; }
    double A = 1.0;
    for (int e = 0; e < E; e++) A *= c;                                         // c^E
    lane_y[t] = carry;
    __syncthreads();
    // backward Hillis-Steele over the 256 runs: Y_t = y_loc_first(t) + c^E Y_{t+1}
    double Y = carry, Ad = A;
    for (int d = 1; d < 256; d <<= 1) {
        const double other = (t + d < 256) ? lane_y[t + d] : 0.0;
        __syncthreads();
        if (t + d < 256) Y = fma(Ad, other, Y);
        lane_y[t] = Y;
        Ad *= Ad;
        __syncthreads();
    }
    const double carry_in = (t < 255) ? lane_y[t + 1] : 0.0;
    double pw = c;
    for (int e = E - 1; e >= 0; e--) { mine[e] = fma(pw, carry_in, mine[e]); pw *= c; }
    __syncthreads();
    double *yo = out + f * n + base;
    for (int i = t; i < len; i += 256) yo[i] = sm[i];
    if (t == 0) heads[f * tiles + T] = sm[0];                                   // the tile's first value, carry-in 0
}

// heads[f][T] -> carry into tile T (the true y at the first sample of tile T + 1)
__global__ void preemphasis_long_carry_kernel(long n_frames, long n, double c, double *__restrict__ heads) {
    const long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const long tiles = (n + PEL_TILE - 1) / PEL_TILE;
    double *h = heads + f * tiles;
    // tile T has len_T samples; y_first(T) = head_loc(T) + c^len_T * y_first(T+1)
    double cp = 1.0;
    for (int e = 0; e < PEL_TILE; e++) cp *= c;                                 // c^PEL_TILE (every tile but the last is full)
    double next = 0.0;                                                          // y at the first sample of the tile after this one
    for (long T = tiles - 1; T >= 0; T--) {
        const double loc = h[T];
        double pw = cp;
        if (T == tiles - 1) { const long len = n - T * PEL_TILE; pw = 1.0; for (long e = 0; e < len; e++) pw *= c; }
        h[T] = next;                                                            // carry into tile T
        next = fma(pw, next, loc);
    }
}

__global__ __launch_bounds__(256) void preemphasis_long_fix_kernel(long n_frames, long n, double c, double *__restrict__ out,
                                                                   const double *__restrict__ heads) {
    const long tiles = (n + PEL_TILE - 1) / PEL_TILE;
    const long f = blockIdx.x / tiles, T = blockIdx.x % tiles;
    if (f >= n_frames) return;
    const double carry = heads[f * tiles + T];
    if (carry == 0.0) return;
    const long base = T * PEL_TILE;
    const int len = (int)((n - base < PEL_TILE) ? n - base : PEL_TILE);
    double *yo = out + f * n + base;
    // y[i] += c^(len - i) carry: lane t owns i = t, t + 256, ... ; c^(len - i) from c^256 steps
    const int t = threadIdx.x;
    double c256 = 1.0;
    for (int e = 0; e < 256; e++) c256 *= c;
    if (t >= len) return;
    int i = t + ((len - 1 - t) / 256) * 256;                                    // the lane's largest i first (smallest power)
    double pw = 1.0;
    for (int e = 0; e < len - i; e++) pw *= c;
    for (; i >= 0; i -= 256) { yo[i] = fma(pw, carry, yo[i]); pw *= c256; }
}

// the reference's sequential order (unstable filters, |c| >= 1)
__global__ void preemphasis_long_serial_kernel(const double *__restrict__ x, long n_frames, long n, long stride, double c,
                                               double *__restrict__ out) {
    const long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const double *xf = x + f * stride;
    double *yo = out + f * n;
    double carry = 0.0;
    for (long i = n - 1; i >= 0; i--) { carry = fma(c, carry, xf[i]); yo[i] = carry; }
}

size_t preemphasis_long_scratch_bytes(long F, long n) { return (size_t)F * (size_t)((n + PEL_TILE - 1) / PEL_TILE) * sizeof(double); }

void launch_preemphasis_long(hipStream_t s, const double *x, long F, long n, long stride, double c, double *out, double *ws) {
    if (!(fabs(c) < 1.0)) {
        hipLaunchKernelGGL(preemphasis_long_serial_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, s, x, F, n, stride, c, out);
        return;
    }
    const long tiles = (n + PEL_TILE - 1) / PEL_TILE;
    hipLaunchKernelGGL(preemphasis_long_local_kernel, dim3((unsigned)(F * tiles)), dim3(256), 0, s, x, F, n, stride, c, out, ws);
    hipLaunchKernelGGL(preemphasis_long_carry_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, s, F, n, c, ws);
    hipLaunchKernelGGL(preemphasis_long_fix_kernel, dim3((unsigned)(F * tiles)), dim3(256), 0, s, F, n, c, out, (const double *)ws);
}

// ---- Pitched::pitch on a long frame ----------------------------------------------------------------------------------------
// r: [F][n] normalised lag sums (Normalize::normalize, src/periodic.rs:404) -> y: [F][2n], (r / w_lag) then the zeros of
// resize(2n, 0) (:406-411)
__global__ void pitch_long_curve_kernel(const double *__restrict__ r, const double *__restrict__ lag_window, long n_frames, long n,
                                        double *__restrict__ y) {
    const long total = n_frames * 2 * n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long f = e / (2 * n), i = e - f * 2 * n;
        y[e] = (i < n) ? r[f * n + i] / lag_window[i] : 0.0;
    }
}

// Peak scan (:413-417, Q4) + parabolic lag (Q5) + frequency filter (:439): the abscissae nn handed to improve_extremum, in
// index order, and their number.  One workgroup of 256 per frame; a tile of 256 lags per step, compacted by a ballot per
// wavefront and a prefix over the four wavefronts.
__global__ __launch_bounds__(256) void pitch_long_peaks_kernel(const double *__restrict__ y, long n_frames, long n, double sample_rate,
                                                               double fmin, double fmax, long cap, double *__restrict__ nn_out,
                                                               int32_t *__restrict__ count) {
    __shared__ int wcount[4];
    const long f = blockIdx.x;
    if (f >= n_frames) return;
    const double *ys = y + f * 2 * n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long b = n / 2;                                                       // brent_ixmax (:414)
    const int offset = (int)(-b - 1);
    long base_out = 0;
    for (long k0 = 0; k0 < b; k0 += 256) {
        const long k = k0 + t;
        bool pass = false;
        double nn = 0.0;
        if (k >= 1 && k + 1 < b) {
            const double c = ys[k];
            if (ys[k - 1] < c && ys[k + 1] < c) {
                double freq;
                cand_from_peak(ys, (int)k, sample_rate, offset, freq, nn);
                pass = (freq == 0.0) || (freq > fmin && freq < fmax);
            }
        }
        const unsigned long long mask = __ballot(pass);
        if (lane == 0) wcount[wave] = __popcll(mask);
        __syncthreads();
        long pos = base_out;
        for (int w = 0; w < wave; w++) pos += wcount[w];
        pos += __popcll(mask & ((1ull << lane) - 1ull));
        if (pass && pos < cap) nn_out[f * cap + pos] = nn;
        base_out += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    if (t == 0) count[f] = (int32_t)(base_out < cap ? base_out : cap);
}

// improve_extremum(.., Sinc(1200), true) (:440-450) per candidate: 16 lanes each; (frequency, strength) into the frame's
// candidate row; flags[f] |= 4 where the reference would index out of bounds, |= 8 for a NaN strength (Q10)
__global__ __launch_bounds__(64) void pitch_long_refine_kernel(const double *__restrict__ y, long n_frames, long n, double sample_rate,
                                                               long cap, const double *__restrict__ nn_in, const int32_t *__restrict__ count,
                                                               double2 *__restrict__ cand, int32_t *__restrict__ flags) {
    const long f = blockIdx.y;
    const long q = (long)blockIdx.x * PNG + lane_id() / PG;
    const int cnt = count[f];
    if ((long)blockIdx.x * PNG >= cnt) return;
    const bool have = q < cnt;
    const double *ys = y + f * 2 * n;
    const long b = n / 2;
    const int offset = (int)(-b - 1), nx = (int)(b - offset), ylen = (int)(2 * n);
    int st = 0;
    double xmid = 0., ymid = 0.;
    improve_extremum_sinc<PG>(ys, ylen, ylen, offset, nx, have ? nn_in[f * cap + q] : 1.0, 1200, have, xmid, ymid, st);
    if (have && (lane_id() & (PG - 1)) == 0) {
#pragma clang fp contract(off)
        double xm = xmid + (double)offset;                                      // :445
        double ym = ymid;
        if (ym > 1.) ym = 1. / ym;                                              // :446
        cand[f * (cap + 1) + q] = double2{sample_rate / xm, ym};                // :447-448
        int fl = (st & 4) ? 4 : 0;
        if (ym != ym) fl |= 8;
        if (fl) atomicOr(&flags[f], fl);
    }
}

// maxima.push(Pitch(0, threshold)); stable sort by strength, descending (:452-453); the first kmax entries, the count, the status
__global__ __launch_bounds__(256) void pitch_long_sort_kernel(long n_frames, long cap, const int32_t *__restrict__ count,
                                                              double2 *__restrict__ cand, const int32_t *__restrict__ flags,
                                                              double threshold, int kmax, double *__restrict__ out_cand, long cand_ld,
                                                              int32_t *__restrict__ out_count, int32_t *__restrict__ status) {
    const long f = blockIdx.x;
    if (f >= n_frames) return;
    const int t = threadIdx.x;
    const int cnt = count[f];
    double2 *row = cand + f * (cap + 1);
    if (t == 0) row[cnt] = double2{0.0, threshold};
    __syncthreads();
    const int total_cand = cnt + 1;
    int code = 0;
    const int fl = flags[f];
    if (fl & 4) code = 4;
    else if (total_cand > 1 && ((fl & 8) || threshold != threshold)) code = 3;   // partial_cmp().unwrap() on NaN (Q10)
    const int total = (code == 0) ? total_cand : 0;
    double *orow = out_cand + f * cand_ld;
    for (int i = t; i < kmax; i += 256) *reinterpret_cast<double2 *>(orow + 2 * i) = double2{0.0, 0.0};
    __syncthreads();
    for (int i = t; i < total; i += 256) {
        const double2 me = row[i];
        int rank = 0;
        for (int j = 0; j < total; j++) { const double sj = row[j].y; rank += (sj > me.y || (sj == me.y && j < i)) ? 1 : 0; }
        if (rank < kmax) *reinterpret_cast<double2 *>(orow + 2 * rank) = me;
    }
    if (t == 0) {
        if (out_count != nullptr) out_count[f] = total;
        if (status != nullptr) status[f] = code;
    }
}

// scratch of one pitch call on F long frames: r [F][n] | y [F][2n] | nn [F][cap] | cand [F][cap + 1][2] | count [F] | flags [F]
static long pitch_long_cap(long n) { return n / 4 + 2; }
size_t pitch_long_scratch_bytes(long F, long n) {
    const long cap = pitch_long_cap(n);
    return ((size_t)F * (size_t)(3 * n + cap + 2 * (cap + 1)) + 2) * sizeof(double) + 2 * (size_t)F * sizeof(int32_t) + 64;
}
static long even_up(long v) { return (v + 1) & ~1L; }
// r must hold Autocorrelate::autocorrelate(n) of every frame, already normalised (the caller runs autocorr_long and the
// row normalisation: they have their own scratch)
void launch_pitch_long(hipStream_t s, long F, long n, const double *lag_window, double sample_rate, double threshold, double fmin,
                       double fmax, int kmax, double *out_cand, long cand_ld, int32_t *out_count, int32_t *status, void *ws) {
    const long cap = pitch_long_cap(n);
    double *r = (double *)ws;                                    // offsets in doubles: r 0 | y F n | nn 3 F n | cand (16-byte aligned)
    double *y = r + F * n;
    double *nn = r + 3 * F * n;
    double2 *cand = reinterpret_cast<double2 *>(r + even_up(3 * F * n + F * cap));
    int32_t *count = reinterpret_cast<int32_t *>(cand + F * (cap + 1));
    int32_t *flags = count + F;
    hipMemsetAsync(flags, 0, (size_t)F * sizeof(int32_t), s);
    long blocks = (F * 2 * n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pitch_long_curve_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const double *)r, lag_window, F, n, y);
    hipLaunchKernelGGL(pitch_long_peaks_kernel, dim3((unsigned)F), dim3(256), 0, s, (const double *)y, F, n, sample_rate, fmin, fmax, cap, nn, count);
    hipLaunchKernelGGL(pitch_long_refine_kernel, dim3((unsigned)((cap + PNG - 1) / PNG), (unsigned)F), dim3(64), 0, s, (const double *)y, F, n,
                       sample_rate, cap, (const double *)nn, (const int32_t *)count, cand, flags);
    hipLaunchKernelGGL(pitch_long_sort_kernel, dim3((unsigned)F), dim3(256), 0, s, F, cap, (const int32_t *)count, cand, (const int32_t *)flags,
                       threshold, kmax, out_cand, cand_ld, out_count, status);
}
double *pitch_long_r(void *ws) { return (double *)ws; }

// ---- MFCC::mfcc on a long frame -----------------------------------------------------------------------------------------
// The bins [b_lo, b_lo + nb) of the n-point DFT by the Goertzel-Reinsch recurrence (k_mfcc.hip: the same constants and steps),
// four bins per lane, one wavefront per 256 bins; the samples reach the lanes as broadcasts of a 64-sample register chunk.
// pu / pd (norm_sqr * slope, norm * slope: src/spectrum.rs:426-434) go to a scratch [F][2][nbp].
constexpr int ML_BPL = 4;
__global__ __launch_bounds__(64) void mfcc_long_bins_kernel(const double *__restrict__ x, long n_frames, long n, long stride,
                                                            const double *__restrict__ window, const double *__restrict__ kappa_sigma,
                                                            const double *__restrict__ slopes, int nb, long nbp, double *__restrict__ pupd) {
    const long f = blockIdx.x;                       // frames on x (no 65,535 cap), blocks of 256 bins on y
    const int lane = lane_id();
    const int p0 = blockIdx.y * 64 * ML_BPL;
    const double *xf = x + f * stride;
    double kap[ML_BPL], sig[ML_BPL], sv[ML_BPL], dv[ML_BPL];
#pragma unroll
    for (int j = 0; j < ML_BPL; j++) {
        const int bi = p0 + j * 64 + lane;
        const bool ok = bi < nb;
        kap[j] = ok ? kappa_sigma[2 * bi] : 0.0;
        sig[j] = ok ? kappa_sigma[2 * bi + 1] : 1.0;
        sv[j] = 0.0; dv[j] = 0.0;
    }
    for (long i0 = 0; i0 < n; i0 += 64) {
        double chunk = 0.0;
        if (i0 + lane < n) { chunk = xf[i0 + lane]; if (window != nullptr) chunk *= window[i0 + lane]; }
        const int steps = (int)((n - i0 < 64) ? n - i0 : 64);
        for (int q = 0; q < steps; q++) {
            const double xi = readlane_f64(chunk, q);
#pragma unroll
            for (int j = 0; j < ML_BPL; j++) {
                const double t = fma(-kap[j], sv[j], dv[j]);
                dv[j] = fma(sig[j], t, xi);
                sv[j] = fma(sig[j], sv[j], dv[j]);
            }
        }
    }
    double *pu = pupd + f * 2 * nbp, *pd = pu + nbp;
#pragma unroll
    for (int j = 0; j < ML_BPL; j++) {
        const int bi = p0 + j * 64 + lane;
        if (bi < nb) {
            const double s2 = sig[j] * (sv[j] - dv[j]);
            double m2 = fma(dv[j], dv[j], sig[j] * kap[j] * sv[j] * s2);
            m2 = (m2 < 0.0) ? 0.0 : m2;
            const double2 sl = *reinterpret_cast<const double2 *>(slopes + 2 * bi);
            pu[bi] = fabs(m2) * sl.x;
            pd[bi] = fabs(sqrt(m2)) * sl.y;
        }
    }
}

__global__ __launch_bounds__(64) void mfcc_long_tail_kernel(long n_frames, const double *__restrict__ pupd, long nbp, const int32_t *__restrict__ bins,
                                                            const double *__restrict__ dct, int num_coeffs, double *__restrict__ out, long out_ld,
                                                            int32_t *__restrict__ status) {
    __shared__ double en[64];
    const long f = blockIdx.x;
    if (f >= n_frames) return;
    const int lane = lane_id();
    const double *pu = pupd + f * 2 * nbp, *pd = pu + nbp;
    mfcc_tail_m(pu, pd, en, bins, dct, num_coeffs, bins[0], lane, out + f * out_ld);
    if (status != nullptr && lane == 0) status[f] = 0;
}

size_t mfcc_long_scratch_bytes(long F, int nb) { return (size_t)F * 2 * (size_t)((nb + 1) & ~1) * sizeof(double); }
void launch_mfcc_long(hipStream_t s, const double *x, long F, long n, long stride, const double *window, const double *kappa_sigma,
                      const int32_t *bins, const double *slopes, const double *dct, int num_coeffs, int nb, double *out, long out_ld,
                      int32_t *status, double *ws) {
    const long nbp = (nb + 1) & ~1;
    hipLaunchKernelGGL(mfcc_long_bins_kernel, dim3((unsigned)F, (unsigned)((nb + 64 * ML_BPL - 1) / (64 * ML_BPL))), dim3(64), 0, s, x, F, n, stride,
                       window, kappa_sigma, slopes, nb, nbp, ws);
    hipLaunchKernelGGL(mfcc_long_tail_kernel, dim3((unsigned)F), dim3(64), 0, s, F, (const double *)ws, nbp, bins, dct, num_coeffs, out, out_ld, status);
}

}  // namespace vbx
