// k_tracker.hip -- EstimateFormants::estimate_formants carried frame to frame
// (src/spectrum.rs:232-333, FormantExtractor :357-369, Q13).
//
// The only cross-frame dependency of the whole hot path: the estimates after frame t feed
// frame t+1 (tests/lib.rs:75-79).  The scan is therefore sequential WITHIN an utterance and
// parallel ACROSS utterances: one lane per segment, segments given by the caller.  All per-lane
// state (6 slots, <= 6 estimates) lives in registers: every loop over slots is fully unrolled
// so no array is ever indexed dynamically (no scratch memory on the serial path).
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

constexpr int NS = VBX_FORMANT_SLOTS_K;   // 6

struct slot_t { bool some; double f, bw; };

__device__ __forceinline__ bool same_res(double f1, double b1, double f2, double b2) {   // derive(PartialEq), :149
    return f1 == f2 && b1 == b2;
}

// comparator of :312-324 (None < Some, then frequency; incomparable -> Equal): a > b ?
__device__ __forceinline__ bool slot_greater(const slot_t &a, const slot_t &b) {
    if (a.some) return b.some ? (a.f > b.f) : true;
    return false;
}

constexpr int TRK_PF = 8;                 // leading row entries prefetched into registers one frame ahead

// One estimate_formants step.  The row has n_res entries of which the first `cnt` are real (the first TRK_PF of
// them arrive in registers, `pre`, the rest are read from memory); the others are the zero padding the reference
// passes along (src/lib.rs:55,114).
// NE: the number of estimates (1..6), a compile-time constant: every loop over estimates has exactly NE trips.
template <int NE>
__device__ __forceinline__ void estimate_formants_step(double (&ef)[NS], double (&eb)[NS],
                                                       const res_t (&pre)[TRK_PF], const res_t *__restrict__ row,
                                                       int n_res, int cnt) {
    constexpr int n_est = NE;
    slot_t s[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) { s[i].some = false; s[i].f = 0.0; s[i].bw = 0.0; }

    // Step 2 (:235-245): nearest resonance per estimate, strict '<' (first wins ties).  One pass over
    // the row serves all estimates.  Entries >= cnt are zeros: the first of them may win, the rest tie.
    {
        double bf[NE], bb[NE], bd[NE];
        const res_t r0 = (cnt > 0) ? pre[0] : res_t{0.0, 0.0};
#pragma unroll
        for (int e = 0; e < NE; e++) { bf[e] = r0.frequency; bb[e] = r0.bandwidth; bd[e] = fabs(r0.frequency - ef[e]); }
        const int lim = (cnt < n_res) ? cnt + 1 : n_res;      // real entries + one representative zero
#pragma unroll
        for (int i = 1; i < TRK_PF; i++) {                    // register-resident entries: static indices
            if (i < lim) {
                const res_t it = (i < cnt) ? pre[i] : res_t{0.0, 0.0};
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const double d = fabs(it.frequency - ef[e]);
                    if (d < bd[e]) { bf[e] = it.frequency; bb[e] = it.bandwidth; bd[e] = d; }
                }
            }
        }
        for (int i = TRK_PF; i < lim; i++) {                  // long rows (orders above 14): from memory
            const res_t it = (i < cnt) ? row[i] : res_t{0.0, 0.0};
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const double d = fabs(it.frequency - ef[e]);
                if (d < bd[e]) { bf[e] = it.frequency; bb[e] = it.bandwidth; bd[e] = d; }
            }
        }
#pragma unroll
        for (int e = 0; e < NE; e++) { s[e].some = true; s[e].f = bf[e]; s[e].bw = bb[e]; }
    }

    // Step 3 (:250-272): w tracks the last surviving slot; all accesses by unrolled select
    bool has_unassigned = false;
    {
        int w = 0;
#pragma unroll
        for (int r = 1; r < NE; r++) {                        // only the first NE slots are occupied here
            if (s[r].some) {
                double wf = 0.0, wb = 0.0, we = 0.0;
#pragma unroll
                for (int q = 0; q < NE; q++) if (q == w) { wf = s[q].f; wb = s[q].bw; we = ef[q]; }
                if (same_res(s[r].f, s[r].bw, wf, wb)) {
                    if (fabs(s[r].f - ef[r]) < fabs(s[r].f - we)) {
#pragma unroll
                        for (int q = 0; q < NE; q++) if (q == w) s[q].some = false;
                        has_unassigned = true; w = r;
                    } else {
                        s[r].some = false; has_unassigned = true;
                    }
                } else {
                    w = r;
                }
            }
        }
    }

    // Step 4 (:274-310).  For j >= 6 no branch of the reference can place a peak, so j stops at 6.
    if (has_unassigned) {
#pragma unroll
        for (int j = 0; j < NS; j++) {
            if (j < n_res) {
                const res_t pk = (j < cnt) ? pre[j] : res_t{0.0, 0.0};         // NS <= TRK_PF
                bool contained = false;
#pragma unroll
                for (int q = 0; q < NS; q++) contained = contained || (s[q].some && same_res(s[q].f, s[q].bw, pk.frequency, pk.bandwidth));
                if (!contained) {
                    if (!s[j].some) {
                        s[j].some = true; s[j].f = pk.frequency; s[j].bw = pk.bandwidth;
                    } else if (j > 0 && !s[j > 0 ? j - 1 : 0].some) {
                        s[j - 1] = s[j];                               // swap(j, j-1); slots[j] = peak
                        s[j].some = true; s[j].f = pk.frequency; s[j].bw = pk.bandwidth;
                    } else if (j + 1 < NS && !s[j + 1 < NS ? j + 1 : j].some) {
                        s[j + 1] = s[j];                               // swap(j, j+1); slots[j] = peak
                        s[j].some = true; s[j].f = pk.frequency; s[j].bw = pk.bandwidth;
                    }
                }
            }
        }
    }

    // :312-324 stable sort (None first, then ascending frequency): adjacent exchanges on strict '>'
#pragma unroll
    for (int pass = 0; pass < NS - 1; pass++) {
#pragma unroll
        for (int i = 0; i + 1 < NS - pass; i++) {
            if (slot_greater(s[i], s[i + 1])) { const slot_t t = s[i]; s[i] = s[i + 1]; s[i + 1] = t; }
        }
    }

    // :327-332 winners with frequency > 0 overwrite the leading estimates
    int e = 0;
#pragma unroll
    for (int i = 0; i < NS; i++) {
        if (s[i].some && s[i].f > 0.0 && e < n_est) {
#pragma unroll
            for (int q = 0; q < NE; q++) if (q == e) { ef[q] = s[i].f; eb[q] = s[i].bw; }
            e++;
        }
    }
}

// Frames [f0 + t0, f0 + t0 + tc) of every segment (t0 = 0, tc = LONG_MAX: whole segments).  A slice that does not start
// a segment continues from the estimates after the previous frame, which are exactly what `out` holds for it -- so the
// scan of a long batch can be cut into time slices that run while the resonances of the next slice are still being
// computed (run_find_formants).
template <int NE>
__global__ __launch_bounds__(64) void tracker_kernel(const res_t *__restrict__ res, long n_frames, int n_res,
                               const int32_t *__restrict__ res_count,
                               const int64_t *__restrict__ seg_start, long n_seg,
                               const res_t *__restrict__ est_init,
                               const int32_t *__restrict__ frame_status, double *__restrict__ out, long out_ld,
                               long t0, long tc) {
    const long sg = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (sg >= n_seg) return;
    const long s0 = (seg_start != nullptr) ? seg_start[sg] : 0;
    const long s1 = (seg_start != nullptr && sg + 1 < n_seg) ? seg_start[sg + 1] : n_frames;
    const long f0 = s0 + t0;
    const long f1 = (s1 - f0 > tc) ? f0 + tc : s1;
    if (f0 >= f1) return;
    double ef[NS], eb[NS];
#pragma unroll
    for (int e = 0; e < NS; e++) { ef[e] = 0.0; eb[e] = 0.0; }
#pragma unroll
    for (int e = 0; e < NE; e++) {
        if (t0 == 0) { ef[e] = est_init[e].frequency; eb[e] = est_init[e].bandwidth; }
        else { const double2 p = *reinterpret_cast<const double2 *>(out + (f0 - 1) * out_ld + 2 * e); ef[e] = p.x; eb[e] = p.y; }
    }
    // The scan is a chain of dependent steps; frame f+1's status, count and leading row entries are requested
    // before frame f is processed so that their latency is off the chain.
    static_assert(NS <= TRK_PF, "step 4 reads its peaks from the prefetched entries");
    res_t cur[TRK_PF];
    int cur_cnt = 0;
    bool cur_ok = false;
    auto fetch = [&](long f, res_t (&dst)[TRK_PF], int &cnt, bool &ok) {
        ok = (frame_status == nullptr) || frame_status[f] == 0;
        cnt = (res_count != nullptr) ? res_count[f] : n_res;
        const res_t *row = res + f * (long)n_res;
#pragma unroll
        for (int i = 0; i < TRK_PF; i++) dst[i] = (i < n_res) ? row[i] : res_t{0.0, 0.0};
    };
    fetch(f0, cur, cur_cnt, cur_ok);
    for (long f = f0; f < f1; f++) {
        res_t nxt[TRK_PF];
        int nxt_cnt = 0;
        bool nxt_ok = false;
        if (f + 1 < f1) fetch(f + 1, nxt, nxt_cnt, nxt_ok);
        if (cur_ok) estimate_formants_step<NE>(ef, eb, cur, res + f * (long)n_res, n_res, cur_cnt);
#pragma unroll
        for (int e = 0; e < NE; e++) { double2 o; o.x = ef[e]; o.y = eb[e]; *reinterpret_cast<double2 *>(out + f * out_ld + 2 * e) = o; }
#pragma unroll
        for (int i = 0; i < TRK_PF; i++) cur[i] = nxt[i];
        cur_cnt = nxt_cnt; cur_ok = nxt_ok;
    }
}

void launch_tracker(hipStream_t s, const res_t *res, long F, int n_res, const int32_t *res_count,
                    const int64_t *seg_start, long n_seg, const res_t *est_init, int n_est,
                    const int32_t *frame_status, res_t *out, long out_ld, long t0, long tc) {
    const int bs = 64;
    const dim3 grid((unsigned)((n_seg + bs - 1) / bs)), block(bs);
    double *o = reinterpret_cast<double *>(out);
#define VBX_TRK(NE) hipLaunchKernelGGL(tracker_kernel<NE>, grid, block, 0, s, res, F, n_res, res_count, seg_start, n_seg, \
                                       est_init, frame_status, o, out_ld, t0, tc)
    switch (n_est) {
        case 1: VBX_TRK(1); break;
        case 2: VBX_TRK(2); break;
        case 3: VBX_TRK(3); break;
        case 4: VBX_TRK(4); break;
        case 5: VBX_TRK(5); break;
        default: VBX_TRK(6); break;
    }
#undef VBX_TRK
}

}  // namespace vbx
