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

    // The usual frame, in every lane of the wave: each estimate kept its own resonance.  Slots 0..NE-1 are Some, the rest
    // None: Step 4 has nothing to place, the stable sort of :312-324 puts the Nones first and orders the NE resonances by
    // frequency (adjacent exchanges on strict '>': the same permutation as the six-slot sort below), and the winners are
    // read off in that order.  A third of the step's instructions were the six-slot sort's flag logic.
    if (!__any(has_unassigned)) {
        double sf[NE], sb[NE];
#pragma unroll
        for (int i = 0; i < NE; i++) { sf[i] = s[i].f; sb[i] = s[i].bw; }
#pragma unroll
        for (int pass = 0; pass < NE - 1; pass++) {
#pragma unroll
            for (int i = 0; i + 1 < NE - pass; i++) {
                if (sf[i] > sf[i + 1]) { const double tf = sf[i], tb = sb[i]; sf[i] = sf[i + 1]; sb[i] = sb[i + 1]; sf[i + 1] = tf; sb[i + 1] = tb; }
            }
        }
        int e = 0;
#pragma unroll
        for (int i = 0; i < NE; i++) {
            if (sf[i] > 0.0) {                                // e < n_est holds: at most NE winners
#pragma unroll
                for (int q = 0; q < NE; q++) if (q == e) { ef[q] = sf[i]; eb[q] = sb[i]; }
                e++;
            }
        }
        return;
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

// ---- the same step on INDICES -------------------------------------------------------------------------------------------
// find_formants hands the tracker rows of a particular shape: `cnt` (<= 6 at order 12) real resonances with strictly
// ascending, positive frequencies, then zeros.  On such a row two entries are equal (derive(PartialEq), :149) exactly when
// they are the same entry -- every zero standing for the one zero entry the reference's strict '<' can ever pick -- so the
// slots can hold entry INDICES instead of (frequency, bandwidth) pairs:
//   Step 2  the nearest entry per estimate: its index and its distance (the same comparisons in the same order);
//   Step 3  "same resonance" is index equality, and the two distances the reference compares (:257-258) are the ones Step 2
//           already holds (both slots hold the same entry);
//   Step 4  contained / placed / swapped on small integers;
//   the sort by frequency is a sort by index (the row ascends; None and the zero entry, whose frequency is not > 0, carry
//   no winner and go last), and the winners are read off the row by index.
// 6-slot select chains on doubles become integer compares and min / max: about half the vector instructions of the
// general step and almost none of its exec-mask branching.  trk_row_qualifies() is the shape test; a wavefront in which any
// lane's row fails it (or any caller without counts: vbx_estimate_formants_f64 on arbitrary rows) takes the general step.
template <int NE>
__device__ __forceinline__ bool trk_row_qualifies(const res_t (&pre)[TRK_PF], int n_res, int cnt) {
    bool q = cnt >= 0 && cnt <= NS && cnt < n_res;            // a zero entry follows the real ones
    double prev = 0.0;
#pragma unroll
    for (int i = 0; i < NS; i++) {
        q = q && (i >= cnt || pre[i].frequency > prev);       // false for NaN
        prev = pre[i].frequency;
    }
    return q;
}

template <int NE>
__device__ __forceinline__ void estimate_formants_step_idx(double (&ef)[NS], double (&eb)[NS], const res_t (&pre)[TRK_PF], int cnt) {
    // entry i of the row as Step 2 sees it: real for i < cnt, the zero entry at i == cnt, nothing beyond
    double fr[NS + 1];
#pragma unroll
    for (int i = 0; i <= NS; i++) fr[i] = (i < cnt && i < TRK_PF) ? pre[i < TRK_PF ? i : 0].frequency : 0.0;
    int sidx[NS];                                             // slot -> entry index
    bool some[NS];
    double bd[NE];
#pragma unroll
    for (int q = 0; q < NS; q++) { sidx[q] = 0; some[q] = q < NE; }
#pragma unroll
    for (int e = 0; e < NE; e++) bd[e] = fabs(fr[0] - ef[e]);
#pragma unroll
    for (int i = 1; i <= NS; i++) {
        const bool valid = i <= cnt;
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const double d = fabs(fr[i] - ef[e]);
            const bool lt = valid && d < bd[e];               // strict '<': the first wins ties
            sidx[e] = lt ? i : sidx[e];
            bd[e] = lt ? d : bd[e];
        }
    }
    // Step 3 (:250-272)
    bool has_unassigned = false;
    {
        int w = 0;
#pragma unroll
        for (int r = 1; r < NE; r++) {
            int wi = 0; double wd = 0.0;
#pragma unroll
            for (int q = 0; q < NE; q++) { wi = (q == w) ? sidx[q] : wi; wd = (q == w) ? bd[q] : wd; }
            const bool same = sidx[r] == wi;
            const bool closer = bd[r] < wd;                   // |v.f - est[r]| < |v.f - est[w]|: both slots hold entry v
#pragma unroll
            for (int q = 0; q < NE; q++) some[q] = some[q] && !(same && closer && q == w) && !(same && !closer && q == r);
            has_unassigned = has_unassigned || same;
            w = (same && !closer) ? w : r;
        }
    }
    // Step 4 (:274-310)
    if (__any(has_unassigned)) {
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const int pj = (j < cnt) ? j : cnt;               // the peak: entry j, or the zero entry
            bool contained = false;
#pragma unroll
            for (int q = 0; q < NS; q++) contained = contained || (some[q] && sidx[q] == pj);
            const bool place = has_unassigned && !contained;
            const bool here = place && !some[j];
            const bool left = place && some[j] && j > 0 && !some[j > 0 ? j - 1 : 0];
            const bool right = place && some[j] && !left && j + 1 < NS && !some[j + 1 < NS ? j + 1 : j];
            if (j > 0) { sidx[j - 1] = left ? sidx[j] : sidx[j - 1]; some[j - 1] = some[j - 1] || left; }
            if (j + 1 < NS) { sidx[j + 1] = right ? sidx[j] : sidx[j + 1]; some[j + 1] = some[j + 1] || right; }
            sidx[j] = (here || left || right) ? pj : sidx[j];
            some[j] = some[j] || here;
        }
    }
    // :312-332: the winners (Some, frequency > 0: a real entry) in ascending frequency = ascending index
    int key[NS];
#pragma unroll
    for (int q = 0; q < NS; q++) key[q] = (some[q] && sidx[q] < cnt) ? sidx[q] : 15;
#define VBX_CE(a, b) { const int lo_ = min(key[a], key[b]), hi_ = max(key[a], key[b]); key[a] = lo_; key[b] = hi_; }
    VBX_CE(0, 1) VBX_CE(2, 3) VBX_CE(4, 5)                    // a 12-exchange network for six keys
    VBX_CE(0, 2) VBX_CE(3, 5) VBX_CE(1, 4)
    VBX_CE(0, 1) VBX_CE(2, 3) VBX_CE(4, 5)
    VBX_CE(1, 2) VBX_CE(3, 4)
    VBX_CE(2, 3)
#undef VBX_CE
#pragma unroll
    for (int e = 0; e < NE; e++) {
        const int k = key[e];
        double nf = ef[e], nb = eb[e];
#pragma unroll
        for (int i = 0; i < NS; i++) { nf = (k == i) ? pre[i].frequency : nf; nb = (k == i) ? pre[i].bandwidth : nb; }
        ef[e] = nf; eb[e] = nb;
    }
}

// one step, by whichever form the wavefront's rows allow.  `general`: VBX_TRACKER_GENERAL=1 (tests: the two forms are
// compared bit for bit).  Must be called from converged code... of the lanes that call it: the ballots see only those.
template <int NE>
__device__ __forceinline__ void estimate_formants_any(double (&ef)[NS], double (&eb)[NS], const res_t (&pre)[TRK_PF],
                                                      const res_t *__restrict__ row, int n_res, int cnt, bool counted, bool general) {
    const bool fast = counted && !general && trk_row_qualifies<NE>(pre, n_res, cnt);
    if (__all(fast)) estimate_formants_step_idx<NE>(ef, eb, pre, cnt);
    else estimate_formants_step<NE>(ef, eb, pre, row, n_res, cnt);
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
                               long t0, long tc, int general) {
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
        if (cur_ok) estimate_formants_any<NE>(ef, eb, cur, res + f * (long)n_res, n_res, cur_cnt, res_count != nullptr, general != 0);
#pragma unroll
        for (int e = 0; e < NE; e++) { double2 o; o.x = ef[e]; o.y = eb[e]; *reinterpret_cast<double2 *>(out + f * out_ld + 2 * e) = o; }
#pragma unroll
        for (int i = 0; i < TRK_PF; i++) cur[i] = nxt[i];
        cur_cnt = nxt_cnt; cur_ok = nxt_ok;
    }
}

// ---- the scan of LONG utterances: speculative chunks with exact repair ------------------------------------------------
// One lane per utterance is one chain of dependent steps, ~5 us each: an hour-long recording tracked as one utterance
// (360,000 frames at a 10 ms hop, what a caller of the reference's find_formants loop over a whole file gets) would take
// two seconds on a single lane.  But the tracker forgets: started from ANY estimates, its state after a few dozen frames is
// bit for bit the state of the true scan (measured on the bench signal: 78 % after 16 frames, 90 % after 32, all after 64
// -- every estimate is overwritten by a resonance of the current frame as soon as that frame has enough of them).  So:
//   A  tracker_spec_kernel    one lane per CHUNK of TRK_CHUNK frames: warm up over the TRK_WARM frames before the chunk
//                             from the initial estimates (exact, not a guess, if an utterance starts inside the warm-up),
//                             then scan the chunk; remember the state the chunk was entered with.
//   B  tracker_check_kernel / tracker_repair_kernel (TRK_ROUNDS times)   every chunk whose remembered entry state is not
//                             the row the previous chunk ended with is redone from that row, in parallel, until its
//                             state meets the row it already holds (from there on the old rows are right).
//   S  tracker_sweep_kernel   one wavefront per utterance walks the chunk boundaries in order and redoes whatever is still
//                             inconsistent: this pass alone makes the result exact (== the sequential scan, bit for bit),
//                             the rounds before it only leave it nothing to do.
#ifndef VBX_TRK_CHUNK
#define VBX_TRK_CHUNK 32
#endif
#ifndef VBX_TRK_WARM
#define VBX_TRK_WARM 32
#endif
#ifndef VBX_TRK_ROUNDS
#define VBX_TRK_ROUNDS 3
#endif
#ifndef VBX_TRK_ROUNDS_LONG
#define VBX_TRK_ROUNDS_LONG 8
#endif
// Check + redo rounds before the sweep.  The sweep redoes what is left SERIALLY, one wavefront per utterance: with thousands of
// utterances in a batch the leftovers of three rounds are spread over thousands of wavefronts (0.3 ms); with ONE utterance
// (round 4: the default of the bench, and what a file is) they queue up in a single wavefront -- which, beside the pipeline's
// analyze kernel, ran 45x slower than alone (a long straight-line step under a thrashed instruction cache): 13.3 ms per
// 1.44 M frames, 52 ms per 4.5 M.  Every further round costs ~0.16 ms and is parallel: after six the sweep took 0.33 ms, after
// ten 8 us (nothing left).  Batches whose utterances average more than 8192 frames take eight rounds.
constexpr int TRK_CHUNK = VBX_TRK_CHUNK, TRK_WARM = VBX_TRK_WARM, TRK_ROUNDS = VBX_TRK_ROUNDS, TRK_ROUNDS_LONG = VBX_TRK_ROUNDS_LONG;

struct trk_in_t {
    const res_t *res; long n_frames; int n_res; const int32_t *res_count; const int64_t *seg_start; long n_seg;
    const res_t *est_init; const int32_t *frame_status; double *out; long out_ld;
    int general;                  // 1: the general step for every frame (VBX_TRACKER_GENERAL)
};
struct trk_spec_t {               // per chunk g
    double *entry;                // [G][2 NS]: the state chunk g's rows were computed from
    double *want;                 // [G][2 NS]: check -> repair: the row the previous chunk ends with
    int32_t *exact;               // [G]: the entry state is not a guess
    int32_t *redo;                // [G]: check -> repair
    int64_t *stop;                // [G]: first utterance start inside the chunk (rows from there on are exact), or the chunk's end
    unsigned long long *mask;     // [ceil(G / 64)]: the redo flags of 64 consecutive chunks as one word (check -> sweep)
};

__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

template <int NE>
__device__ __forceinline__ void trk_init(const trk_in_t &in, double (&ef)[NS], double (&eb)[NS]) {
#pragma unroll
    for (int e = 0; e < NS; e++) { ef[e] = 0.0; eb[e] = 0.0; }
#pragma unroll
    for (int e = 0; e < NE; e++) { ef[e] = in.est_init[e].frequency; eb[e] = in.est_init[e].bandwidth; }
}
template <int NE>
__device__ __forceinline__ void trk_load_row(const trk_in_t &in, long f, double (&ef)[NS], double (&eb)[NS]) {
#pragma unroll
    for (int e = 0; e < NS; e++) { ef[e] = 0.0; eb[e] = 0.0; }
#pragma unroll
    for (int e = 0; e < NE; e++) { const double2 p = *reinterpret_cast<const double2 *>(in.out + f * in.out_ld + 2 * e); ef[e] = p.x; eb[e] = p.y; }
}
template <int NE>
__device__ __forceinline__ void trk_store_row(const trk_in_t &in, long f, const double (&ef)[NS], const double (&eb)[NS]) {
#pragma unroll
    for (int e = 0; e < NE; e++) { double2 o; o.x = ef[e]; o.y = eb[e]; *reinterpret_cast<double2 *>(in.out + f * in.out_ld + 2 * e) = o; }
}
template <int NE>
__device__ __forceinline__ bool trk_row_is(const trk_in_t &in, long f, const double (&ef)[NS], const double (&eb)[NS]) {
    bool same = true;
#pragma unroll
    for (int e = 0; e < NE; e++) {
        const double2 p = *reinterpret_cast<const double2 *>(in.out + f * in.out_ld + 2 * e);
        same = same && same_bits(p.x, ef[e]) && same_bits(p.y, eb[e]);
    }
    return same;
}
// one frame of the scan (a frame whose status is not 0 leaves the estimates untouched, src/lib.rs:75)
template <int NE>
__device__ __forceinline__ void trk_frame(const trk_in_t &in, long f, double (&ef)[NS], double (&eb)[NS]) {
    const bool ok = (in.frame_status == nullptr) || in.frame_status[f] == 0;
    if (!ok) return;
    const int cnt = (in.res_count != nullptr) ? in.res_count[f] : in.n_res;
    const res_t *row = in.res + f * (long)in.n_res;
    res_t pre[TRK_PF];
#pragma unroll
    for (int i = 0; i < TRK_PF; i++) pre[i] = (i < in.n_res) ? row[i] : res_t{0.0, 0.0};
    estimate_formants_any<NE>(ef, eb, pre, row, in.n_res, cnt, in.res_count != nullptr, in.general != 0);
}
// The chunked scan's steps with the NEXT frame's status, count and leading row entries requested before the current frame is
// processed (round 6; tracker_kernel above always did): a chunk is a chain of 64 dependent steps, and every step of rounds 4-5
// began by waiting for its own row to arrive from the L2 / HBM -- the lane's wavefront is alone on its SIMD (1 M frames are
// 488 wavefronts), nothing else covers that wait.  Same operations on the same values.
struct trk_row_t { res_t pre[TRK_PF]; int cnt; bool ok; };
__device__ __forceinline__ void trk_fetch(const trk_in_t &in, long f, trk_row_t &r) {
    r.ok = (in.frame_status == nullptr) || in.frame_status[f] == 0;
    r.cnt = (in.res_count != nullptr) ? in.res_count[f] : in.n_res;
    const res_t *row = in.res + f * (long)in.n_res;
#pragma unroll
    for (int i = 0; i < TRK_PF; i++) r.pre[i] = (i < in.n_res) ? row[i] : res_t{0.0, 0.0};
}
template <int NE>
__device__ __forceinline__ void trk_frame_pre(const trk_in_t &in, long f, const trk_row_t &r, double (&ef)[NS], double (&eb)[NS]) {
    if (!r.ok) return;                                        // a frame whose status is not 0 leaves the estimates untouched (src/lib.rs:75)
    estimate_formants_any<NE>(ef, eb, r.pre, in.res + f * (long)in.n_res, in.n_res, r.cnt, in.res_count != nullptr, in.general != 0);
}
// index of the utterance that holds frame f, and the frame at which the next one starts
__device__ __forceinline__ long trk_segment_of(const trk_in_t &in, long f, long &next_start) {
    if (in.seg_start == nullptr) { next_start = in.n_frames; return 0; }
    long lo = 0, hi = in.n_seg - 1;                           // seg_start[0] == 0 <= f
    while (lo < hi) { const long mid = (lo + hi + 1) >> 1; if (in.seg_start[mid] <= f) lo = mid; else hi = mid - 1; }
    next_start = (lo + 1 < in.n_seg) ? in.seg_start[lo + 1] : in.n_frames;
    return lo;
}

template <int NE>
__global__ __launch_bounds__(64) void tracker_spec_kernel(const trk_in_t in, const trk_spec_t sp, long n_chunks) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_chunks) return;
    const long f_begin = g * TRK_CHUNK;
    const long f_end = (f_begin + TRK_CHUNK < in.n_frames) ? f_begin + TRK_CHUNK : in.n_frames;
    const long w_begin = (f_begin > TRK_WARM) ? f_begin - TRK_WARM : 0;
    long next_start;
    long sg = trk_segment_of(in, w_begin, next_start);
    const long seg_begin = (in.seg_start != nullptr) ? in.seg_start[sg] : 0;
    bool exact = (w_begin == seg_begin);
    long stop = f_end;
    double ef[NS], eb[NS];
    trk_init<NE>(in, ef, eb);
    trk_row_t cur;
    trk_fetch(in, w_begin, cur);
    for (long f = w_begin; f < f_end; f++) {
        trk_row_t nxt = cur;
        if (f + 1 < f_end) trk_fetch(in, f + 1, nxt);
        if (f == next_start) {                                 // an utterance starts here: the state is known
            trk_init<NE>(in, ef, eb);
            exact = true;
            if (f > f_begin && stop == f_end) stop = f;
            do {                                               // (empty utterances: several starts on one frame)
                sg++;
                next_start = (sg + 1 < in.n_seg) ? in.seg_start[sg + 1] : in.n_frames;
            } while (next_start == f);
        }
        if (f == f_begin) {
#pragma unroll
            for (int e = 0; e < NS; e++) { sp.entry[g * 2 * NS + 2 * e] = ef[e]; sp.entry[g * 2 * NS + 2 * e + 1] = eb[e]; }
            sp.exact[g] = exact ? 1 : 0;
        }
        trk_frame_pre<NE>(in, f, cur, ef, eb);
        if (f >= f_begin) trk_store_row<NE>(in, f, ef, eb);
        cur = nxt;
    }
    sp.stop[g] = stop;
}

// does chunk g's entry state equal the row the previous chunk ends with?  (reads rows only: no lane writes here)
template <int NE>
__global__ __launch_bounds__(64) void tracker_check_kernel(const trk_in_t in, const trk_spec_t sp, long n_chunks) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int redo = 0;
    if (g < n_chunks && g > 0 && !sp.exact[g]) {
        double tf[NS], tb[NS];
        trk_load_row<NE>(in, g * TRK_CHUNK - 1, tf, tb);
        bool same = true;
#pragma unroll
        for (int e = 0; e < NE; e++) same = same && same_bits(tf[e], sp.entry[g * 2 * NS + 2 * e]) && same_bits(tb[e], sp.entry[g * 2 * NS + 2 * e + 1]);
        if (!same) {
            redo = 1;
#pragma unroll
            for (int e = 0; e < NS; e++) { sp.want[g * 2 * NS + 2 * e] = tf[e]; sp.want[g * 2 * NS + 2 * e + 1] = tb[e]; }
        }
    }
    if (g < n_chunks) sp.redo[g] = redo;
    // the same flags, 64 chunks to a word (this wavefront's chunks are [64 blockIdx.x, 64 blockIdx.x + 64)): the sweep reads
    // 4096 chunks per step instead of 64 -- a 12.5-hour utterance has 140,000 chunks, and one wavefront walks them all
    const unsigned long long m = __ballot(redo != 0);
    if (threadIdx.x == 0) sp.mask[blockIdx.x] = m;
}

// redo the flagged chunks from the state the check saw, each inside its own rows
template <int NE>
__global__ __launch_bounds__(64) void tracker_repair_kernel(const trk_in_t in, const trk_spec_t sp, long n_chunks) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_chunks || !sp.redo[g]) return;
    double ef[NS], eb[NS];
#pragma unroll
    for (int e = 0; e < NS; e++) {
        ef[e] = sp.want[g * 2 * NS + 2 * e]; eb[e] = sp.want[g * 2 * NS + 2 * e + 1];
        sp.entry[g * 2 * NS + 2 * e] = ef[e]; sp.entry[g * 2 * NS + 2 * e + 1] = eb[e];      // what the rows now follow from
    }
    const long stop = sp.stop[g];
    if (g * TRK_CHUNK >= stop) return;
    trk_row_t cur;
    trk_fetch(in, g * TRK_CHUNK, cur);
    for (long f = g * TRK_CHUNK; f < stop; f++) {
        trk_row_t nxt = cur;
        if (f + 1 < stop) trk_fetch(in, f + 1, nxt);           // (a redo usually ends after a few frames: one row requested in vain)
        trk_frame_pre<NE>(in, f, cur, ef, eb);
        if (trk_row_is<NE>(in, f, ef, eb)) break;              // met the old scan: its remaining rows stand
        trk_store_row<NE>(in, f, ef, eb);
        cur = nxt;
    }
}

// the guarantee: utterance by utterance, chunk boundary by chunk boundary, in order.  One WAVEFRONT per utterance: the lanes
// read 64 of the check kernel's flags at a time (a 10-hour utterance has 56,000 chunks); lane 0 redoes a flagged chunk from
// the row before it until its state meets rows that follow from it.  Rows past that point, and therefore the flags of the
// chunks past it, are unchanged; the boundaries the redo crosses are compared on the way.
template <int NE>
__global__ __launch_bounds__(64) void tracker_sweep_kernel(const trk_in_t in, const trk_spec_t sp) {
    const long sg = (long)blockIdx.x;
    const int lane = (int)threadIdx.x;
    const long s0 = (in.seg_start != nullptr) ? in.seg_start[sg] : 0;
    const long s1 = (in.seg_start != nullptr && sg + 1 < in.n_seg) ? in.seg_start[sg + 1] : in.n_frames;
    const long g_end = (s1 + TRK_CHUNK - 1) / TRK_CHUNK;       // chunks g with g * TRK_CHUNK < s1
    long g = s0 / TRK_CHUNK + 1;                               // the first chunk that starts inside the utterance
    while (g < g_end) {
        // lane l looks at the word of chunks [64 (g / 64 + l), + 64), cut to [g, g_end): 4096 chunks per step
        const long w0 = g >> 6, wi = w0 + lane;
        unsigned long long word = (wi * 64 < g_end) ? sp.mask[wi] : 0ull;
        if (wi == w0) word &= ~0ull << (g & 63);
        if ((wi + 1) * 64 > g_end) { const long keep = g_end - wi * 64; word &= (keep <= 0) ? 0ull : ((keep >= 64) ? ~0ull : ((1ull << keep) - 1ull)); }
        const unsigned long long any = __ballot(word != 0ull);
        if (any == 0ull) { g = (w0 + 64) * 64; continue; }
        const int wl = __builtin_ctzll(any);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(word & 0xffffffffull), wl);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(word >> 32), wl);
        const unsigned long long mask = ((unsigned long long)hi << 32) | lo;
        g = (w0 + wl) * 64;                                    // `first` below = g + ctz(mask)
        const long first = g + __builtin_ctzll(mask);
        long f = first * TRK_CHUNK;
        if (lane == 0) {
            double ef[NS], eb[NS];
            trk_load_row<NE>(in, f - 1, ef, eb);
            bool redo = false;
#pragma unroll
            for (int e = 0; e < NE; e++) redo = redo || !same_bits(ef[e], sp.entry[first * 2 * NS + 2 * e]) || !same_bits(eb[e], sp.entry[first * 2 * NS + 2 * e + 1]);
            if (redo) {
                while (f < s1) {
                    if (f > first * TRK_CHUNK && f % TRK_CHUNK == 0) {   // entering the next chunk: its rows follow from sp.entry
                        const long g2 = f / TRK_CHUNK;
                        bool same = true;
#pragma unroll
                        for (int e = 0; e < NE; e++) same = same && same_bits(ef[e], sp.entry[g2 * 2 * NS + 2 * e]) && same_bits(eb[e], sp.entry[g2 * 2 * NS + 2 * e + 1]);
                        if (same) break;
                    }
                    trk_frame<NE>(in, f, ef, eb);
                    if (trk_row_is<NE>(in, f, ef, eb)) { f++; break; }
                    trk_store_row<NE>(in, f, ef, eb);
                    f++;
                }
            }
        }
        f = __shfl(f, 0, 64);
        const long after = (f + TRK_CHUNK - 1) / TRK_CHUNK;     // the first boundary at or after the frame the redo ended on
        g = (after > first + 1) ? after : first + 1;            // (a boundary the redo stopped ON was accepted there)
    }
}

// ---- a track that continues ACROSS two launches (a recording sharded over GPUs by frame ranges) ---------------
// The shard's rows [first, stop) were tracked from a GUESS of the state after frame first - 1 (the rank warmed its tracker up
// over the `first` frames before its range, starting from the initial estimates).  state_in is the TRUE state: the formant row
// the previous shard ends with.  If it is, bit for bit, the row this shard holds at first - 1, every later row follows from it
// and stands.  Otherwise the scan is redone from the true state until it meets a row it already holds (the tracker forgets:
// a few dozen frames) -- the repair step of the chunked scan, with the previous rank in the role of the previous chunk.
// changed: number of rows rewritten (0: the guess was right).  One lane: a chain of dependent steps.
template <int NE>
__global__ __launch_bounds__(64) void tracker_stitch_kernel(const trk_in_t in, long first, long stop,
                                                            const double *__restrict__ state_in, int32_t *__restrict__ changed) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double ef[NS], eb[NS];
#pragma unroll
    for (int e = 0; e < NS; e++) { ef[e] = 0.0; eb[e] = 0.0; }
#pragma unroll
    for (int e = 0; e < NE; e++) { ef[e] = state_in[2 * e]; eb[e] = state_in[2 * e + 1]; }
    int n = 0;
    if (!(first > 0 && trk_row_is<NE>(in, first - 1, ef, eb))) {
        for (long f = first; f < stop; f++) {
            trk_frame<NE>(in, f, ef, eb);
            if (trk_row_is<NE>(in, f, ef, eb)) break;          // met the rows already there: the rest follows from them
            trk_store_row<NE>(in, f, ef, eb);
            n++;
        }
    }
    if (changed != nullptr) *changed = n;
}

// VBX_TRACKER_GENERAL=1 (read per launch): the general step on every frame instead of the index form
static int tracker_general() { const char *e = getenv("VBX_TRACKER_GENERAL"); return (e && atoi(e) != 0) ? 1 : 0; }

size_t tracker_chunked_workspace_bytes(long F) {
    const size_t G = (size_t)((F + TRK_CHUNK - 1) / TRK_CHUNK);
    return G * (2 * 2 * NS * sizeof(double) + 2 * sizeof(int32_t) + sizeof(int64_t)) + ((G + 63) / 64 + 1) * sizeof(unsigned long long) + 64;
}

// The same result as launch_tracker (whole segments), for batches with long utterances.  ws: tracker_chunked_workspace_bytes(F).
void launch_tracker_chunked(hipStream_t s, const res_t *res, long F, int n_res, const int32_t *res_count,
                            const int64_t *seg_start, long n_seg, const res_t *est_init, int n_est,
                            const int32_t *frame_status, res_t *out, long out_ld, void *ws) {
    const long G = (F + TRK_CHUNK - 1) / TRK_CHUNK;
    trk_in_t in{res, F, n_res, res_count, seg_start, seg_start != nullptr ? n_seg : 1, est_init, frame_status,
                reinterpret_cast<double *>(out), out_ld, tracker_general()};
    trk_spec_t sp;
    char *w = reinterpret_cast<char *>(ws);
    sp.entry = reinterpret_cast<double *>(w); w += (size_t)G * 2 * NS * sizeof(double);
    sp.want = reinterpret_cast<double *>(w); w += (size_t)G * 2 * NS * sizeof(double);
    sp.stop = reinterpret_cast<int64_t *>(w); w += (size_t)G * sizeof(int64_t);
    sp.mask = reinterpret_cast<unsigned long long *>(w); w += (size_t)((G + 63) / 64 + 1) * sizeof(unsigned long long);
    sp.exact = reinterpret_cast<int32_t *>(w); w += (size_t)G * sizeof(int32_t);
    sp.redo = reinterpret_cast<int32_t *>(w);
    const dim3 block(64), grid_c((unsigned)((G + 63) / 64)), grid_s((unsigned)in.n_seg);
    const int rounds = (F / (in.n_seg > 0 ? in.n_seg : 1) > 8192) ? TRK_ROUNDS_LONG : TRK_ROUNDS;
#define VBX_TRK_CH(NE)                                                                             \
    do {                                                                                           \
        hipLaunchKernelGGL(tracker_spec_kernel<NE>, grid_c, block, 0, s, in, sp, G);               \
        for (int r = 0; r < rounds; r++) {                                                         \
            hipLaunchKernelGGL(tracker_check_kernel<NE>, grid_c, block, 0, s, in, sp, G);          \
            hipLaunchKernelGGL(tracker_repair_kernel<NE>, grid_c, block, 0, s, in, sp, G);         \
        }                                                                                          \
        hipLaunchKernelGGL(tracker_check_kernel<NE>, grid_c, block, 0, s, in, sp, G);              \
        hipLaunchKernelGGL(tracker_sweep_kernel<NE>, grid_s, block, 0, s, in, sp);                 \
    } while (0)
    switch (n_est) {
        case 1: VBX_TRK_CH(1); break;
        case 2: VBX_TRK_CH(2); break;
        case 3: VBX_TRK_CH(3); break;
        case 4: VBX_TRK_CH(4); break;
        case 5: VBX_TRK_CH(5); break;
        default: VBX_TRK_CH(6); break;
    }
#undef VBX_TRK_CH
}

void launch_tracker_stitch(hipStream_t s, const res_t *res, long F, int n_res, const int32_t *res_count, int n_est,
                           const int32_t *frame_status, res_t *out, long out_ld, long first, long stop,
                           const double *state_in, int32_t *changed) {
    trk_in_t in{res, F, n_res, res_count, nullptr, 1, nullptr, frame_status, reinterpret_cast<double *>(out), out_ld, tracker_general()};
#define VBX_TRK_ST(NE) hipLaunchKernelGGL(tracker_stitch_kernel<NE>, dim3(1), dim3(64), 0, s, in, first, stop, state_in, changed)
    switch (n_est) {
        case 1: VBX_TRK_ST(1); break;
        case 2: VBX_TRK_ST(2); break;
        case 3: VBX_TRK_ST(3); break;
        case 4: VBX_TRK_ST(4); break;
        case 5: VBX_TRK_ST(5); break;
        default: VBX_TRK_ST(6); break;
    }
#undef VBX_TRK_ST
}

void launch_tracker(hipStream_t s, const res_t *res, long F, int n_res, const int32_t *res_count,
                    const int64_t *seg_start, long n_seg, const res_t *est_init, int n_est,
                    const int32_t *frame_status, res_t *out, long out_ld, long t0, long tc) {
    const int bs = 64;
    const dim3 grid((unsigned)((n_seg + bs - 1) / bs)), block(bs);
    double *o = reinterpret_cast<double *>(out);
#define VBX_TRK(NE) hipLaunchKernelGGL(tracker_kernel<NE>, grid, block, 0, s, res, F, n_res, res_count, seg_start, n_seg, \
                                       est_init, frame_status, o, out_ld, t0, tc, tracker_general())
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
