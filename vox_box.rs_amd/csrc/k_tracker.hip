// k_tracker.hip -- EstimateFormants::estimate_formants carried frame to frame
// (src/spectrum.rs:232-333, FormantExtractor :357-369, Q13).
//
// The only cross-frame dependency of the whole hot path: the estimates after frame t feed
// frame t+1 (tests/lib.rs:75-79).  The scan is therefore sequential WITHIN an utterance and
// parallel ACROSS utterances: one lane per segment, segments given by the caller.
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

struct opt_res { int some; res_t v; };

__device__ __forceinline__ bool res_eq(const res_t &a, const res_t &b) {   // derive(PartialEq), :149
    return a.frequency == b.frequency && a.bandwidth == b.bandwidth;
}
__device__ __forceinline__ bool slots_contains(const opt_res *slots, const res_t &peak) {
    for (int i = 0; i < VBX_FORMANT_SLOTS_K; i++) if (slots[i].some && res_eq(slots[i].v, peak)) return true;
    return false;
}
// comparator of :312-324 (None < Some, then frequency; incomparable -> Equal)
__device__ __forceinline__ int slot_cmp(const opt_res &a, const opt_res &b) {
    if (a.some) {
        if (b.some) {
            if (a.v.frequency < b.v.frequency) return -1;
            if (a.v.frequency > b.v.frequency) return 1;
            return 0;
        }
        return 1;
    }
    return -1;
}

// one estimate_formants step.  res row has n_res entries of which the first `cnt` are read
// from memory and the rest are the zero padding the reference passes along (src/lib.rs:55,114).
__device__ void estimate_formants_step(res_t *est, int n_est, const res_t *__restrict__ row, int n_res, int cnt) {
    opt_res slots[VBX_FORMANT_SLOTS_K];
    for (int i = 0; i < VBX_FORMANT_SLOTS_K; i++) { slots[i].some = 0; slots[i].v.frequency = 0.0; slots[i].v.bandwidth = 0.0; }
    const res_t zero = {0.0, 0.0};
    const int n_zip = n_est < VBX_FORMANT_SLOTS_K ? n_est : VBX_FORMANT_SLOTS_K;

    // Step 2 (:235-245)
    for (int e = 0; e < n_zip; e++) {
        res_t best = (cnt > 0) ? row[0] : zero;
        double bestd = fabs(best.frequency - est[e].frequency);
        for (int i = 1; i < n_res; i++) {
            const res_t it = (i < cnt) ? row[i] : zero;
            const double d = fabs(it.frequency - est[e].frequency);
            if (d < bestd) { best = it; bestd = d; }
            if (i >= cnt) break;        // all further entries equal this zero: strict '<' never fires again
        }
        slots[e].some = 1; slots[e].v = best;
    }

    // Step 3 (:250-272)
    int w = 0;
    bool has_unassigned = false;
    for (int r = 1; r < VBX_FORMANT_SLOTS_K; r++) {
        if (!slots[r].some) continue;
        const res_t v = slots[r].v;
        if (res_eq(v, slots[w].v)) {
            if (fabs(v.frequency - est[r].frequency) < fabs(v.frequency - est[w].frequency)) {
                slots[w].some = 0; has_unassigned = true; w = r;
            } else {
                slots[r].some = 0; has_unassigned = true;
            }
        } else {
            w = r;
        }
    }

    // Step 4 (:274-310)
    if (has_unassigned) {
        for (int j = 0; j < n_res; j++) {
            const res_t peak = (j < cnt) ? row[j] : zero;
            if (slots_contains(slots, peak)) {
                if (j >= cnt) break;    // once the zero peak is contained it stays contained
                continue;
            }
            if (j < VBX_FORMANT_SLOTS_K && !slots[j].some) { slots[j].some = 1; slots[j].v = peak; continue; }
            if (j > 0 && j < VBX_FORMANT_SLOTS_K) {
                if (!slots[j - 1].some) {
                    const opt_res t = slots[j]; slots[j] = slots[j - 1]; slots[j - 1] = t;
                    slots[j].some = 1; slots[j].v = peak; continue;
                }
            }
            if (j + 1 < VBX_FORMANT_SLOTS_K && !slots[j + 1].some) {
                const opt_res t = slots[j]; slots[j] = slots[j + 1]; slots[j + 1] = t;
                slots[j].some = 1; slots[j].v = peak; continue;
            }
            if (j >= VBX_FORMANT_SLOTS_K && j >= cnt) break;   // zero peaks beyond the slots can never be placed
        }
    }

    // :312-324 stable sort
    for (int i = 1; i < VBX_FORMANT_SLOTS_K; i++) {
        const opt_res key = slots[i]; int j = i;
        while (j > 0 && slot_cmp(slots[j - 1], key) > 0) { slots[j] = slots[j - 1]; j--; }
        slots[j] = key;
    }
    // :327-332
    int e = 0;
    for (int i = 0; i < VBX_FORMANT_SLOTS_K && e < n_est; i++)
        if (slots[i].some && slots[i].v.frequency > 0.0) est[e++] = slots[i].v;
}

__global__ void tracker_kernel(const res_t *__restrict__ res, long n_frames, int n_res,
                               const int32_t *__restrict__ res_count,
                               const int64_t *__restrict__ seg_start, long n_seg,
                               const res_t *__restrict__ est_init, int n_est,
                               const int32_t *__restrict__ frame_status, res_t *__restrict__ out) {
    const long sg = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (sg >= n_seg) return;
    const long f0 = (seg_start != nullptr) ? seg_start[sg] : 0;
    const long f1 = (seg_start != nullptr && sg + 1 < n_seg) ? seg_start[sg + 1] : n_frames;
    res_t est[VBX_FORMANT_SLOTS_K];
    for (int e = 0; e < n_est; e++) est[e] = est_init[e];
    for (long f = f0; f < f1; f++) {
        const bool ok = (frame_status == nullptr) || frame_status[f] == 0;
        if (ok) {
            const int cnt = (res_count != nullptr) ? res_count[f] : n_res;
            estimate_formants_step(est, n_est, res + f * (long)n_res, n_res, cnt);
        }
        for (int e = 0; e < n_est; e++) out[f * (long)n_est + e] = est[e];
    }
}

void launch_tracker(hipStream_t s, const res_t *res, long F, int n_res, const int32_t *res_count,
                    const int64_t *seg_start, long n_seg, const res_t *est_init, int n_est,
                    const int32_t *frame_status, res_t *out) {
    const int bs = 64;
    hipLaunchKernelGGL(tracker_kernel, dim3((unsigned)((n_seg + bs - 1) / bs)), dim3(bs), 0, s,
                       res, F, n_res, res_count, seg_start, n_seg, est_init, n_est, frame_status, out);
}

}  // namespace vbx
