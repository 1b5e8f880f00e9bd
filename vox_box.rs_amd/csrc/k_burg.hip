// k_burg.hip -- Burg LPC (LPC::lpc_praat_mut, src/spectrum.rs:101-146, Q12).
//
// G lanes per frame, 64/G frames per wavefront.  A lane keeps b1/b2 elements
// [lig*EPL, (lig+1)*EPL) of its frame in registers, so the reference's shift
//   b2[j] <- b2[j+1] - a*b1[j+1]
// needs ONE neighbour-lane fetch per order (DPP wave_shl) instead of a pass through memory.
// Per order: two group reductions (num, denum; DPP, bit-identical inside the group), the
// coefficient recursion with coefficient t in lane t of the group (the reversed operand aa[i-2-t] by one
// lane permute: no LDS, no serial loop), one fused update sweep.
// The shrinking valid range [0, N-i) is kept by zeroing exactly the element that drops out.
// Short frames use small groups so that the per-order overhead is shared by several frames.
//
// The denominator.  The reference sums b1^2 + b2^2 over the valid range at every order (:118-121).  With mu = 2 num / den
// the updated arrays satisfy, exactly in real arithmetic,
//     den' = (1 - mu^2) den - (b1[last] - mu b2[last])^2 - (b2[0] - mu b1[0])^2
// (the two elements that leave the range: `last` = the one the update zeroes, and the one the shift drops at the front),
// which costs two broadcasts instead of 2 EPL FMAs and a group reduction per order -- a quarter of the kernel's vector
// instructions, and the kernel is vector-issue bound.  Its rounding differs from the direct sums' by ~eps / (1 - mu^2)
// per order (relative), so it is used only while that stays far inside the 1e-6 coefficient tolerance: after an order with
// 1 - mu^2 < 2^-20, or once den' has fallen below 2^-24 of the first order's den (absolute errors ~eps * den_1 would show),
// the next order sums directly again.  The status test `den <= 0` (:123-125) can only fire on a directly summed
// denominator: a recursion value that small has already handed over to the direct sums.  Only the one-frame-per-wavefront
// form (G = 64: frames of more than 1024 samples) does this: there the choice is a scalar branch that depends on the frame
// alone (a frame's result must not depend on which frames share its wavefront), and that is where it pays (measured:
// Burg at N = 1200 23.4 -> 19.5 ms per 4.5 M frames; at N = 512 with four frames per wavefront the permutes cost what the
// FMAs save).
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

// T: Sample type of the frames and of the coefficients (double; float = the f32 instantiation: widened on load, the
// windowed product rounded to T first, the recursion in f64, one rounding to T on the store).
// TIN: what the frames hold when it is not T: int16_t = 16-bit PCM (widened in registers, s / 32767; window and results f64).
template <int G, int EPL, typename T, typename TIN = T>
__global__ __launch_bounds__(64) void burg_kernel(
    const TIN *__restrict__ x, long n_frames, int n, long stride, const T *__restrict__ window,
    int p, T *__restrict__ out, int32_t *__restrict__ status, const frame_map_t map,
    const int32_t *__restrict__ list = nullptr, const int32_t *__restrict__ list_count = nullptr) {
    constexpr int NG = 64 / G;
    static_assert(G == 16 || G == 32 || G == 64, "one coefficient per lane of the group: orders up to G (the launchers choose)");
    const int lane = lane_id();
    const int gid = lane / G, lig = lane % G;
    // one-frame workgroups of a hop-strided view: neighbouring frames on the same XCD (vbx_device.hpp, xcd_item)
    const long blk = (NG == 1 && list == nullptr) ? xcd_item(blockIdx.x, gridDim.x) : (long)blockIdx.x;
    // list != nullptr: the frames named by list[0 .. *list_count) (the ones k_burg_fast.hip's guard turned away), a fixed
    // grid striding over a count only the device knows
    for (long it = blk * NG;; it += (long)gridDim.x * NG) {
    long f;
    if (list != nullptr) {
        const long cnt = *list_count;
        if (it >= cnt) break;
        f = (it + gid < cnt) ? (long)list[it + gid] : -1;
    } else f = frame_map(map, it + gid, n_frames);
    const bool have = f >= 0;
    const TIN *xf = x + (have ? f : 0) * stride;
    constexpr bool PCM = sizeof(TIN) == 2;

    double b1[EPL], b2[EPL];
    // A lane's EPL samples are contiguous: 16-byte loads where the lane lies inside the frame and the rows are aligned
    // (one double at a time, the 64 lanes of an instruction touch 64 different cache lines EPL times over: the address
    // path, not the arithmetic, then bounds the kernel -- measured 2.72 -> 1.84 ms per million 512-sample frames).
    bool vec = false;
    if constexpr (sizeof(T) == 8 && !PCM && EPL % 2 == 0) {
        vec = have && (lig + 1) * EPL <= n && ((((uintptr_t)xf) | ((uintptr_t)window)) & 15) == 0;
        if (vec) {
            const double2 *xv = reinterpret_cast<const double2 *>(xf + lig * EPL);
            const double2 *wv = reinterpret_cast<const double2 *>(window != nullptr ? window + lig * EPL : xf + lig * EPL);
#pragma unroll
            for (int e = 0; e < EPL; e += 2) {
                const double2 v = xv[e / 2];
                if (window != nullptr) { const double2 w = wv[e / 2]; b1[e] = v.x * w.x; b1[e + 1] = v.y * w.y; }
                else { b1[e] = v.x; b1[e + 1] = v.y; }
            }
        }
    }
    if (!vec && (!have || lig * EPL >= n)) {                 // a lane past the frame (or without one): zeros, no loads
#pragma unroll
        for (int e = 0; e < EPL; e++) b1[e] = 0.0;
    } else if (!vec) {                                       // a lane that straddles the frame's end, unaligned rows, floats
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int j = lig * EPL + e;
            double v = (have && j < n) ? (PCM ? pcm16_value((int)xf[j]) : (double)xf[j]) : 0.0;
            if (window != nullptr && j < n) v = (double)(T)(v * (double)window[j]);
            b1[e] = v;
        }
    }
    const bool last_lane = (lig == G - 1);          // its "next lane" belongs to another frame
    // b2[j] = x[j+1]  (zero past the frame);  b1[j] = x[j] for j <= n-2  (src/spectrum.rs:108-114)
    {
        const double fetched = from_next_lane(b1[0]);   // DPP outside any lane-dependent branch
        const double nxt = last_lane ? 0.0 : fetched;
#pragma unroll
        for (int e = 0; e < EPL - 1; e++) b2[e] = b1[e + 1];
        b2[EPL - 1] = nxt;
        // the slot index is the same in every lane: a scalar compare per slot, one masked move for the slot that matches
        // (as a per-lane select chain this cost 4 vector instructions per slot and order)
        const int last = n - 1;
        const int kb = __builtin_amdgcn_readfirstlane(last % EPL), lb = last / EPL;
#pragma unroll
        for (int e = 0; e < EPL; e++) if (e == kb) { asm volatile("" : "+v"(b1[e])); if (lig == lb) b1[e] = 0.0; }   // the empty asm pins the branch
    }

    int st = 0;
    double aa = 0.0, co = 0.0;                       // lane t of the group: aa[t], coeffs[t]  (src/spectrum.rs:116-139)
    const int gbase = lane - lig;
    constexpr bool DEN_RECURSION = (G == 64);
    bool den_known = false;                          // den of this order follows from the previous order (wave-uniform)
    double den_next = 0.0, den_first = 0.0;
    for (int i = 1; i <= p; i++) {
        // independent accumulators (even / odd slots; b1^2 and b2^2 apart): one chain of 3 EPL dependent FMAs was the
        // latency of the whole order
        double num0 = 0.0, num1 = 0.0;
#pragma unroll
        for (int e = 0; e + 1 < EPL; e += 2) {
            num0 = fma(b1[e], b2[e], num0);
            num1 = fma(b1[e + 1], b2[e + 1], num1);
        }
        if (EPL & 1) num0 = fma(b1[EPL - 1], b2[EPL - 1], num0);
        double num = group_sum<G>(num0 + num1), den;
        if (den_known) den = den_next;
        else {
            double da0 = 0.0, da1 = 0.0, db0 = 0.0, db1 = 0.0;
#pragma unroll
            for (int e = 0; e + 1 < EPL; e += 2) {
                da0 = fma(b1[e], b1[e], da0);
                da1 = fma(b1[e + 1], b1[e + 1], da1);
                db0 = fma(b2[e], b2[e], db0);
                db1 = fma(b2[e + 1], b2[e + 1], db1);
            }
            if (EPL & 1) { da0 = fma(b1[EPL - 1], b1[EPL - 1], da0); db0 = fma(b2[EPL - 1], b2[EPL - 1], db0); }
            den = group_sum<G>((da0 + da1) + (db0 + db1));
            if (i == 1) den_first = den;
        }
        if (st == 0 && den <= 0.0) st = 1;           // Err(LPC), src/spectrum.rs:123-125 (NaN falls through)
        const double c = 2.0 * num / den;
        {   // coeffs[i-1] = c;  coeffs[j-1] = aa[j-1] - c * aa[i-j-1], j = 1..i-1   (t = j-1 <-> lane t)
            int srcl = i - 2 - lig;
            srcl = (srcl < 0) ? 0 : srcl;
            const double rev = __shfl(aa, gbase + srcl, 64);
            if (lig < i - 1) co = aa - c * rev;
            else if (lig == i - 1) co = c;
        }
        if (i < p) {
            if (lig < i) aa = co;                    // aa[j-1] = coeffs[j-1], j = 1..i
            const double a = c;                      // aa[i-1] == coeffs[i-1]
            const double f1 = from_next_lane(b1[0]), f2 = from_next_lane(b2[0]);
            const double nb1 = last_lane ? 0.0 : f1;
            const double nb2 = last_lane ? 0.0 : f2;
            const double e_front = fma(-a, b1[0], b2[0]);    // lane 0 of the group: the element the shift drops, b2[0] - mu b1[0]
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const double b1n = (e + 1 < EPL) ? b1[e + 1] : nb1;   // old b1[j+1]
                const double b2n = (e + 1 < EPL) ? b2[e + 1] : nb2;   // old b2[j+1]
                const double t1 = fma(-a, b2[e], b1[e]);
                const double t2 = fma(-a, b1n, b2n);
                b1[e] = t1;
                b2[e] = t2;
            }
            // element n-i-1 leaves the valid range (the update loop runs j-1 < n-i-1)
            const int drop = n - i - 1;
            den_known = false;
            if (drop >= 0) {
                const int kb = __builtin_amdgcn_readfirstlane(drop % EPL), lb = drop / EPL;
                double e_back = 0.0;                 // lane lb of the group: b1[last] - mu b2[last], just computed
#pragma unroll
                for (int e = 0; e < EPL; e++)
                    if (e == kb) { asm volatile("" : "+v"(b1[e]), "+v"(b2[e])); e_back = b1[e]; if (lig == lb) { b1[e] = 0.0; b2[e] = 0.0; } }
                if constexpr (DEN_RECURSION) {
                    // the next order's denominator from this one (header comment); the two dropped elements by one
                    // lane broadcast each (lb is a scalar)
                    const double eb = readlane_f64(e_back, __builtin_amdgcn_readfirstlane(lb)), ef = readlane_f64(e_front, 0);
                    const double omm = fma(-a, a, 1.0);
                    den_next = fma(-ef, ef, fma(-eb, eb, omm * den));
                    const bool fine = omm > 0x1p-20 && den_next > den_first * 0x1p-24;     // NaN: not fine
                    den_known = __builtin_amdgcn_readfirstlane((int)fine) != 0;            // identical in every lane
                }
            }
        }
    }
    if (have) {
        if (lig < p) out[f * (long)p + lig] = (T)((st == 0) ? co * -1.0 : 0.0);   // :142-144
        if (status != nullptr && lig == 0) status[f] = st;
    }
    if (list == nullptr) break;
    }
}

bool burg_supported(int n, int p) {
    return n >= 2 && n <= 64 * 64 && p >= 1 && p <= VBX_MAX_LPC_ORDER_K;
}

// orders above 16 need more than 16 lanes per frame (one coefficient per lane), orders above 32 all 64
static bool burg_small_groups_ok(int p) { return p <= 16; }
static bool burg_half_wave_ok(int p) { return p <= 32; }

template <typename T>
static void launch_burg_t(hipStream_t s, const T *x, long F, int n, long stride, const T *window,
                          int p, T *out, int32_t *status, frame_map_t map) {
    dim3 b(64);
    const long items = frame_map_items(map, F);
#define VBX_BURG(GG, E)                                                                                          \
    hipLaunchKernelGGL((burg_kernel<GG, E, T>), dim3((unsigned)((items + (64 / GG) - 1) / (64 / GG))), b, 0, s, \
                       x, F, n, stride, window, p, out, status, map)
    const bool g16 = burg_small_groups_ok(p);
    if (g16 && n <= 16 * 8) VBX_BURG(16, 8);
    else if (g16 && n <= 16 * 16) VBX_BURG(16, 16);
    else if (g16 && n <= 16 * 32) VBX_BURG(16, 32);
    else if (n <= 32 * 32 && burg_half_wave_ok(p)) VBX_BURG(32, 32);
    else if (n <= 64 * 20) VBX_BURG(64, 20);
    else if (n <= 64 * 32) VBX_BURG(64, 32);
    else VBX_BURG(64, 64);
#undef VBX_BURG
}

void launch_burg(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                 int p, double *out, int32_t *status, frame_map_t map) {
    launch_burg_t<double>(s, x, F, n, stride, window, p, out, status, map);
}
void launch_burg_pcm16(hipStream_t s, const int16_t *x, long F, int n, long stride, const double *window,
                       int p, double *out, int32_t *status, frame_map_t map) {
    dim3 b(64);
    const long items = frame_map_items(map, F);
#define VBX_BURG16(GG, E)                                                                                                  \
    hipLaunchKernelGGL((burg_kernel<GG, E, double, int16_t>), dim3((unsigned)((items + (64 / GG) - 1) / (64 / GG))), b, 0, s, \
                       x, F, n, stride, window, p, out, status, map)
    const bool g16 = burg_small_groups_ok(p);
    if (g16 && n <= 16 * 32) VBX_BURG16(16, 32);
    else if (n <= 32 * 32 && burg_half_wave_ok(p)) VBX_BURG16(32, 32);
    else if (n <= 64 * 20) VBX_BURG16(64, 20);
    else if (n <= 64 * 32) VBX_BURG16(64, 32);
    else VBX_BURG16(64, 64);
#undef VBX_BURG16
}
// the direct recursion on the frames of a device-side list (k_burg_fast.hip)
template <typename TIN>
static void launch_burg_list_t(hipStream_t s, const TIN *x, long F, int n, long stride, const double *window,
                               int p, double *out, int32_t *status, const int32_t *list, const int32_t *count) {
    dim3 b(64);
    const frame_map_t map{0, 0, 0};
    const long cap = 8192;                                   // wavefronts (8 per SIMD); each strides over the list
#define VBX_BURGL(GG, E)                                                                                                   \
    hipLaunchKernelGGL((burg_kernel<GG, E, double, TIN>), dim3((unsigned)((F + (64 / GG) - 1) / (64 / GG) < cap ? (F + (64 / GG) - 1) / (64 / GG) : cap)), b, 0, s, \
                       x, F, n, stride, window, p, out, status, map, list, count)
    const bool g16 = burg_small_groups_ok(p);
    if (g16 && n <= 16 * 32) VBX_BURGL(16, 32);
    else if (n <= 32 * 32 && burg_half_wave_ok(p)) VBX_BURGL(32, 32);
    else if (n <= 64 * 20) VBX_BURGL(64, 20);
    else if (n <= 64 * 32) VBX_BURGL(64, 32);
    else VBX_BURGL(64, 64);
#undef VBX_BURGL
}
void launch_burg_list(hipStream_t s, const double *x, long F, int n, long stride, const double *window,
                      int p, double *out, int32_t *status, const int32_t *list, const int32_t *count) {
    launch_burg_list_t<double>(s, x, F, n, stride, window, p, out, status, list, count);
}
void launch_burg_pcm16_list(hipStream_t s, const int16_t *x, long F, int n, long stride, const double *window,
                            int p, double *out, int32_t *status, const int32_t *list, const int32_t *count) {
    launch_burg_list_t<int16_t>(s, x, F, n, stride, window, p, out, status, list, count);
}

void launch_burg_f32(hipStream_t s, const float *x, long F, int n, long stride, const float *window,
                     int p, float *out, int32_t *status) {
    launch_burg_t<float>(s, x, F, n, stride, window, p, out, status, frame_map_t{0, 0, 0});
}

}  // namespace vbx
