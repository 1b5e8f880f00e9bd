// vbx_host.cpp -- every entry point of the ABI that is plain host arithmetic: the window / lag-window tables by the sample
// crate's recurrences, the mel bins, the work-size and frame-count formulas, and the geometry of a recording sharded over
// ranks (frame ranges, warm-up, gather transfer list).  No HIP here: the same file is built into libvoxbox_hip.so and, under
// -fsanitize=address,undefined, into a host-only library that tests/test_sanitizers.py drives with random arguments
// (tools/host_asan/Makefile).
#include "../../include/voxbox_hip.h"
#include "vbx_host.hpp"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

extern "C" int vbx_internal_fail(vbx_ctx *ctx, int code, const char *msg);   // vbx_api.hip (or the sanitizer build's stub)

namespace {
int fail(vbx_ctx *ctx, int code, const std::string &msg) { return vbx_internal_fail(ctx, code, msg.c_str()); }
}  // namespace

namespace vbx {

// sample 0.10 signal::Phase: yields phase, then phase = (phase + step) % 1.0
static void phase_ramp(std::vector<double> &ph, size_t n, double step) {
    ph.resize(n);
    double next = 0.0;
    for (size_t i = 0; i < n; i++) { ph[i] = next; next = std::fmod(next + step, 1.0); }
}

int window_table_host(int kind, size_t n, double *out) {
    const double pi2 = M_PI * 2.0;
    std::vector<double> ph;
    switch (kind) {
        case VBX_WINDOW_HANNING:            // Window::<Hanning>::new(n)
            phase_ramp(ph, n, 1.0 / ((double)n - 1.0));
            for (size_t i = 0; i < n; i++) out[i] = 0.5 * (1.0 - std::cos(ph[i] * pi2));
            return VBX_SUCCESS;
        case VBX_WINDOW_HANNING_LAG:        // HanningLag::at_phase, src/periodic.rs:239-247 (Q3)
            phase_ramp(ph, n, 1.0 / ((double)n - 1.0));
            for (size_t i = 0; i < n; i++) {
                const double v = ph[i] * pi2;
                out[i] = (1.0 - ph[i]) * (2.0 / 3.0 + (1.0 / 3.0) * std::cos(v)) + (1.0 / pi2) * std::sin(v);
            }
            return VBX_SUCCESS;
        case VBX_WINDOW_HANNING_PERIODIC: { // src/lib.rs:65-70
            const double len_inv = 1.0 / (double)n;
            for (size_t i = 0; i < n; i++) out[i] = 0.5 * (1.0 - std::cos(((double)i * len_inv) * pi2));
            return VBX_SUCCESS;
        }
        case VBX_WINDOW_RECTANGLE:
            for (size_t i = 0; i < n; i++) out[i] = 1.0;
            return VBX_SUCCESS;
    }
    return VBX_E_INVALID;
}

// src/spectrum.rs:411-414 (Q14)
void mel_bins_host(size_t n, size_t k, double lo, double hi, double sr, std::vector<int32_t> &bins, bool &overflow) {
    const double mlo = vbx_hz_to_mel(lo), mel_range = vbx_hz_to_mel(hi) - mlo;
    bins.resize(k + 2);
    overflow = false;
    for (size_t i = 0; i < k + 2; i++) {
        const double point = ((double)i / (double)k) * mel_range + mlo;
        const double b = std::floor((double)(n + 1) * vbx_mel_to_hz(point) / sr);
        if (!(b >= 0.0)) { bins[i] = 0; }
        else if (b > 1.0e9) { bins[i] = 1000000000; overflow = true; }
        else bins[i] = (int32_t)b;
    }
}


}  // namespace vbx

using namespace vbx;

extern "C" {

int vbx_window_table_f64(int kind, size_t n, double *h_out) {
    if (!h_out || n < 1) return fail(nullptr, VBX_E_INVALID, "vbx_window_table_f64: bad argument");
    int rc = window_table_host(kind, n, h_out);
    if (rc != VBX_SUCCESS) return fail(nullptr, rc, "vbx_window_table_f64: unknown window kind");
    return rc;
}

size_t vbx_frame_count(size_t n_samples, size_t frame_len, size_t hop) {
    if (frame_len == 0 || hop == 0 || n_samples < frame_len) return 0;
    return (n_samples - frame_len) / hop + 1;
}

double vbx_hz_to_mel(double hz) { return 1125. * std::log1p(hz / 700.); }       // src/spectrum.rs:375-377
double vbx_mel_to_hz(double mel) { return 700. * (std::exp(mel / 1125.) - 1.); } // src/spectrum.rs:379-381
size_t vbx_find_formants_real_work_size(size_t buf_len, size_t n_coeffs) { return buf_len * 2 + n_coeffs * 23 + 2; }
size_t vbx_find_formants_complex_work_size(size_t n_coeffs) { return n_coeffs * 7 + 4; }

// The mel filter bank's bins (src/spectrum.rs:411-414: K + 2 mel points, floor((N + 1) hz / sr), Q14).  Returns 1 when the
// reference would panic on every frame of this geometry (a bin beyond the spectrum, or descending bins: usize underflow).
int vbx_mfcc_bins(size_t frame_len, size_t num_coeffs, double lo_hz, double hi_hz, double sample_rate, int32_t *h_bins) {
    if (!h_bins || frame_len < 1 || num_coeffs < 1) return fail(nullptr, VBX_E_INVALID, "vbx_mfcc_bins: bad argument");
    std::vector<int32_t> b; bool bad = false;
    mel_bins_host(frame_len, num_coeffs, lo_hz, hi_hz, sample_rate, b, bad);
    for (size_t i = 0; i + 1 < b.size(); i++) if (b[i + 1] < b[i]) bad = true;
    if (b.back() > (int32_t)(frame_len < 0x7fffffff ? frame_len : 0x7fffffff)) bad = true;
    std::memcpy(h_bins, b.data(), b.size() * sizeof(int32_t));
    return bad ? 1 : VBX_SUCCESS;
}

int vbx_window_table_f32(int kind, size_t n, float *h_out) {
    if (!h_out || n == 0) return fail(nullptr, VBX_E_INVALID, "vbx_window_table_f32: bad argument");
    std::vector<double> t(n);
    int rc = vbx_window_table_f64(kind, n, t.data());
    if (rc != VBX_SUCCESS) return rc;
    for (size_t i = 0; i < n; i++) h_out[i] = (float)t[i];
    return VBX_SUCCESS;
}

size_t vbx_degree_c64(const vbx_complex *h_poly, size_t len) {          // src/polynomial.rs:26-28
    if (!h_poly) return 0;
    for (size_t i = len; i-- > 0;) if (!(h_poly[i].re == 0.0 && h_poly[i].im == 0.0)) return i;
    return 0;
}
size_t vbx_off_low_c64(const vbx_complex *h_poly, size_t len) {         // src/polynomial.rs:30-32
    if (!h_poly) return 0;
    for (size_t i = 0; i < len; i++) if (!(h_poly[i].re == 0.0 && h_poly[i].im == 0.0)) return i;
    return 0;
}

size_t vbx_resampled_len(size_t frame_len, double resample_ratio) {
    return (size_t)std::ceil(resample_ratio * (double)frame_len);          // src/lib.rs:42
}

size_t vbx_record_doubles(const vbx_analysis_params *h_p) {
    if (!h_p) return 0;
    size_t n = 2;                                                   // Pitch { frequency, strength }
    if (h_p->formant_order) n += 2 * h_p->n_est;                    // Resonance { frequency, bandwidth } x n_est
    if (h_p->mfcc_coeffs) n += h_p->mfcc_coeffs;
    if (h_p->lpc_order) n += h_p->lpc_order + 1;
    return n;
}

int vbx_gather_plan(const int64_t *h_rows, int world, int rank, int dst, size_t row_doubles,
                    int64_t *h_offset, int64_t *h_count, int32_t *h_op) {
    if (!h_rows || world < 1 || rank < 0 || rank >= world || dst < 0 || dst >= world || row_doubles < 1)
        return fail(nullptr, VBX_E_INVALID, "vbx_gather_plan: bad argument");
    int64_t off = 0;
    for (int r = 0; r < world; r++) {
        if (h_rows[r] < 0) return fail(nullptr, VBX_E_INVALID, "vbx_gather_plan: negative row count");
        const int64_t cnt = h_rows[r] * (int64_t)row_doubles;
        if (h_offset) h_offset[r] = off;
        if (h_count) h_count[r] = cnt;
        if (h_op) {
            int32_t op = VBX_GATHER_NONE;
            if (rank == dst) { if (cnt > 0) op = (r == dst) ? VBX_GATHER_COPY : VBX_GATHER_RECV; }
            else if (r == dst && h_rows[rank] > 0) op = VBX_GATHER_SEND;       // whatever dst itself contributes
            h_op[r] = op;
        }
        off += cnt;
    }
    return VBX_SUCCESS;
}

// last utterance start <= f (0 without a segment list)
static size_t seg_start_of(const int64_t *h_seg_start, size_t n_segments, size_t f) {
    size_t best = 0;
    if (h_seg_start) for (size_t i = 0; i < n_segments; i++) { if ((size_t)h_seg_start[i] <= f) best = (size_t)h_seg_start[i]; else break; }
    return best;
}
// first utterance start > f, or n_frames
static size_t seg_start_after(const int64_t *h_seg_start, size_t n_segments, size_t f, size_t n_frames) {
    if (h_seg_start) for (size_t i = 0; i < n_segments; i++) if ((size_t)h_seg_start[i] > f) return (size_t)h_seg_start[i] < n_frames ? (size_t)h_seg_start[i] : n_frames;
    return n_frames;
}

int vbx_shard_range(size_t n_frames, int world, int rank, const int64_t *h_seg_start, size_t n_segments,
                    size_t *lo, size_t *hi) {
    if (!lo || !hi || world < 1 || rank < 0 || rank >= world) return fail(nullptr, VBX_E_INVALID, "vbx_shard_range: bad argument");
    auto even_hi = [&](int r) {                      // end of rank r under the plain even split
        const size_t base = n_frames / (size_t)world, rem = n_frames % (size_t)world;
        return (size_t)(r + 1) * base + ((size_t)(r + 1) < rem ? (size_t)(r + 1) : rem);
    };
    // The even cut, unless an utterance starts within 1/32 of a shard after it: a rank that begins where an utterance
    // begins needs nothing from its predecessor.  A cut INSIDE an utterance is fine too -- the track is carried across
    // it (vbx_shard_plan, vbx_comm_stitch_tracks_f64) -- so one long utterance splits evenly.
    const size_t slack = n_frames / (size_t)world / 32;
    auto cut = [&](int r) -> size_t {
        if (r < 0) return 0;
        if (r >= world - 1) return n_frames;
        const size_t target = even_hi(r);
        if (!h_seg_start || n_segments == 0) return target;
        for (size_t i = 0; i < n_segments; i++) {    // first boundary >= target
            const size_t b = (size_t)h_seg_start[i];
            if (b >= target) return (b <= target + slack && b <= n_frames) ? b : target;
        }
        return target;
    };
    size_t a = cut(rank - 1), b = cut(rank);
    if (b < a) b = a;
    *lo = a; *hi = b;
    return VBX_SUCCESS;
}

int vbx_shard_plan(size_t n_frames, int world, int rank, const int64_t *h_seg_start, size_t n_segments, vbx_shard_plan_t *out) {
    if (!out) return fail(nullptr, VBX_E_INVALID, "vbx_shard_plan: null output");
    if (h_seg_start && n_segments > 0) {
        if (h_seg_start[0] != 0) return fail(nullptr, VBX_E_INVALID, "vbx_shard_plan: seg_start[0] must be 0");
        for (size_t i = 1; i < n_segments; i++)
            if (h_seg_start[i] < h_seg_start[i - 1]) return fail(nullptr, VBX_E_INVALID, "vbx_shard_plan: seg_start must ascend");
    }
    size_t lo = 0, hi = 0;
    int rc = vbx_shard_range(n_frames, world, rank, h_seg_start, n_segments, &lo, &hi);
    if (rc != VBX_SUCCESS) return rc;
    // does the utterance that holds frame `c` reach back further than a warm-up can cover exactly?
    auto continued = [&](size_t c) { return c > 0 && c < n_frames && c - seg_start_of(h_seg_start, n_segments, c) > (size_t)VBX_SHARD_WARM_FRAMES; };
    out->lo = lo; out->hi = hi;
    out->warm = 0; out->stop = 0; out->continues_prev = 0; out->continues_next = 0;
    if (hi <= lo) return VBX_SUCCESS;                // an empty shard (more ranks than frames): nothing to do, nothing to pass on
    const size_t back = lo - seg_start_of(h_seg_start, n_segments, lo);
    out->warm = back < (size_t)VBX_SHARD_WARM_FRAMES ? back : (size_t)VBX_SHARD_WARM_FRAMES;
    out->continues_prev = continued(lo) ? 1 : 0;
    out->continues_next = continued(hi) ? 1 : 0;
    const size_t next = seg_start_after(h_seg_start, n_segments, lo, n_frames);
    out->stop = ((next < hi) ? next : hi) - (lo - out->warm);
    return VBX_SUCCESS;
}

int vbx_shard_local_segments(const vbx_shard_plan_t *h_plan, const int64_t *h_seg_start, size_t n_segments,
                             int64_t *h_out, size_t cap, size_t *n_out) {
    if (!h_plan || !n_out) return fail(nullptr, VBX_E_INVALID, "vbx_shard_local_segments: null argument");
    const size_t first = h_plan->lo - h_plan->warm;
    size_t n = 0;
    if (h_out && n < cap) h_out[n] = 0;
    n++;
    if (h_seg_start) for (size_t i = 0; i < n_segments; i++) {
        const size_t b = (size_t)h_seg_start[i];
        if (b > first && b < h_plan->hi) { if (h_out && n < cap) h_out[n] = (int64_t)(b - first); n++; }
    }
    *n_out = n;
    if (h_out && n > cap) return fail(nullptr, VBX_E_INVALID, "vbx_shard_local_segments: output too small");
    return VBX_SUCCESS;
}

int vbx_shard_samples(size_t lo, size_t hi, size_t frame_len, size_t hop, size_t *s0, size_t *s1) {
    if (!s0 || !s1 || frame_len < 1 || hop < 1) return fail(nullptr, VBX_E_INVALID, "vbx_shard_samples: bad argument");
    *s0 = lo * hop;
    *s1 = (hi <= lo) ? lo * hop : (hi - 1) * hop + frame_len;     // includes the frame_len - hop halo
    return VBX_SUCCESS;
}


}  // extern "C"
