// voxbox.hpp -- header-only C++17 mirror of the vox_box 0.3.0 trait surface over the C ABI
// (include/voxbox_hip.h).  The reference is compiled code (Rust), so the host side above the
// ABI is C++: the same names, argument meaning and error behaviour as the crate's traits, with
// one difference of kind -- every call takes a BATCH of frames (the user's frame loop,
// examples/pitch_detection.rs:23-30, tests/lib.rs:71-83) instead of one slice.
//
//   reference (per frame)                              here (per batch)
//   frame.autocorrelate(n)            periodic.rs:265  Autocorrelate::autocorrelate(ctx, frames, n)
//   r.normalize()                     waves.rs:60      Normalize::normalize(ctx, rows, F, n)
//   r.lpc(p)                          spectrum.rs:86   LPC::lpc(ctx, r, F, r_stride, p)
//   frame.lpc_praat(p)                spectrum.rs:94   LPC::lpc_praat(ctx, frames, p, status)
//   frame.pitch::<Hanning>(..)        periodic.rs:356  Pitched::pitch(ctx, frames, sr, thr, min, max, kmax, ..)
//   poly.find_roots_mut(work)         polynomial.rs:92 Polynomial::find_roots_mut(ctx, polys, F, len, status)
//   roots.to_resonance(sr)            spectrum.rs:204  ToResonance::to_resonance(ctx, roots, F, n, sr, ..)
//   est.estimate_formants(res)        spectrum.rs:232  EstimateFormants::estimate_formants(ctx, ..) (= FormantExtractor)
//   frame.mfcc(k, (lo, hi), sr)       spectrum.rs:410  MFCC::mfcc(ctx, frames, k, lo, hi, sr, ..)
//   vox_box::find_formants(..)        lib.rs:40        find_formants(ctx, frames, sr, p, segments, est, ..)
//
// VoxBoxResult<()> / panics of a frame become a per-frame status code (Status), API misuse and
// runtime failures throw voxbox::Error.  There is no CPU fallback.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "voxbox_hip.h"

namespace voxbox {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// VoxBoxError (src/error.rs:4-16) + the panics of the path, per frame
enum class Status : int32_t { Ok = 0, LPC = 1, Polynomial = 2, NaN = 3, Panic = 4 };

using Resonance = vbx_resonance;   // #[repr(C)] Resonance<f64>, spectrum.rs:149-154
using Pitch = vbx_pitch;           // Pitch<f64>, periodic.rs:306-310
using Complex = vbx_complex;       // num::Complex<f64>

constexpr size_t MAX_RESONANCES = VBX_MAX_RESONANCES;                       // lib.rs:26
inline const double *male_formant_estimates() { return VBX_MALE_FORMANT_ESTIMATES; }     // lib.rs:27
inline const double *female_formant_estimates() { return VBX_FEMALE_FORMANT_ESTIMATES; } // lib.rs:28

class Context {
public:
    explicit Context(int device = 0, void *hip_stream = nullptr) {
        int rc = vbx_ctx_create(&ctx_, device, hip_stream);
        if (rc != VBX_SUCCESS) throw Error(rc, vbx_last_error(nullptr));
    }
    ~Context() { vbx_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    vbx_ctx *get() const { return ctx_; }
    void check(int rc) const { if (rc != VBX_SUCCESS) throw Error(rc, vbx_last_error(ctx_)); }
    void sync() const { check(vbx_sync(ctx_)); }
private:
    vbx_ctx *ctx_ = nullptr;
};

// device allocation owned by the caller (the crate's `Vec` returning variants allocate; the
// `_mut` variants below take caller-owned device memory and never allocate)
template <typename T>
class DeviceVec {
public:
    DeviceVec(Context &c, size_t n) : c_(&c), n_(n) { c.check(vbx_malloc(c.get(), &p_, n * sizeof(T))); }
    DeviceVec(Context &c, const std::vector<T> &h) : DeviceVec(c, h.size()) {
        c.check(vbx_memcpy_h2d(c.get(), p_, h.data(), h.size() * sizeof(T)));
    }
    DeviceVec(DeviceVec &&o) noexcept : c_(o.c_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
    DeviceVec(const DeviceVec &) = delete;
    ~DeviceVec() { if (p_) vbx_free(c_->get(), p_); }
    T *data() const { return static_cast<T *>(p_); }
    size_t size() const { return n_; }
    std::vector<T> to_host() const {
        std::vector<T> h(n_);
        c_->check(vbx_memcpy_d2h(c_->get(), h.data(), p_, n_ * sizeof(T)));
        return h;
    }
private:
    Context *c_;
    void *p_ = nullptr;
    size_t n_;
};

// sample::window tables on the host (sample 0.10 recurrences), upload with DeviceVec
enum class Window : int { Hanning = VBX_WINDOW_HANNING, HanningLag = VBX_WINDOW_HANNING_LAG,
                          HanningPeriodic = VBX_WINDOW_HANNING_PERIODIC, Rectangle = VBX_WINDOW_RECTANGLE };
inline std::vector<double> window_table(Window w, size_t n) {
    std::vector<double> t(n);
    if (vbx_window_table_f64(static_cast<int>(w), n, t.data()) != VBX_SUCCESS) throw Error(VBX_E_INVALID, "window_table");
    return t;
}

// A batch of frames in device memory: Windower::{hanning,rectangle}(samples, bin, hop) as a view
// (stride = hop) or a dense [F, N] array (stride = N).  `window` is the Windower's multiplier table.
struct Frames {
    const double *x = nullptr;
    size_t n_frames = 0, frame_len = 0, stride = 0;
    const double *window = nullptr;
    static Frames windower(const double *samples, size_t n_samples, size_t bin, size_t hop, const double *window) {
        return Frames{samples, vbx_frame_count(n_samples, bin, hop), bin, hop, window};
    }
    static Frames dense(const double *x, size_t n_frames, size_t frame_len, const double *window = nullptr) {
        return Frames{x, n_frames, frame_len, frame_len, window};
    }
};

struct Autocorrelate {   // periodic.rs:265-274
    static void autocorrelate_mut(Context &c, const Frames &f, size_t n_coeffs, double *coeffs /* [F, n_coeffs] */) {
        c.check(vbx_autocorrelate_f64(c.get(), f.x, f.n_frames, f.frame_len, f.stride, f.window, n_coeffs, coeffs));
    }
    static DeviceVec<double> autocorrelate(Context &c, const Frames &f, size_t n_coeffs) {
        DeviceVec<double> out(c, f.n_frames * n_coeffs);
        autocorrelate_mut(c, f, n_coeffs, out.data());
        return out;
    }
    // impl Autocorrelate for VecDeque (periodic.rs:291-304): the Windower view over a ring buffer as a dense batch
    static DeviceVec<double> ring_frames(Context &c, const double *ring, size_t capacity, size_t head, size_t n_frames,
                                         size_t frame_len, size_t stride) {
        DeviceVec<double> out(c, n_frames * frame_len);
        c.check(vbx_ring_frames_f64(c.get(), ring, capacity, head, n_frames, frame_len, stride, out.data()));
        return out;
    }
};

struct RMS {             // waves.rs:10-23
    static void rms(Context &c, const Frames &f, double *out /* [F] */) {
        c.check(vbx_rms_f64(c.get(), f.x, f.n_frames, f.frame_len, f.stride, f.window, out));
    }
};

struct Filter {          // waves.rs:82-96 (result in a dense [F, N] batch: strided frames overlap)
    static void preemphasis(Context &c, const Frames &f, double factor, double *out) {
        c.check(vbx_preemphasis_f64(c.get(), f.x, f.n_frames, f.frame_len, f.stride, factor, out));
    }
};

// hound-style ingestion of 16-bit PCM (tests/lib.rs:17-19): sample / 32767
inline void pcm16_to_f64(Context &c, const int16_t *pcm, size_t n, double *out) { c.check(vbx_pcm16_to_f64(c.get(), pcm, n, out)); }

struct Normalize {       // waves.rs:60-76
    static void normalize(Context &c, double *rows, size_t n_rows, size_t n) { c.check(vbx_normalize_f64(c.get(), rows, n_rows, n)); }
};

struct LPC {             // spectrum.rs:50-55
    static void lpc_mut(Context &c, const double *r, size_t n_frames, size_t r_stride, size_t n_coeffs, double *ac) {
        c.check(vbx_lpc_f64(c.get(), r, n_frames, r_stride, n_coeffs, ac));
    }
    static DeviceVec<double> lpc(Context &c, const double *r, size_t n_frames, size_t r_stride, size_t n_coeffs) {
        DeviceVec<double> out(c, n_frames * (n_coeffs + 1));
        lpc_mut(c, r, n_frames, r_stride, n_coeffs, out.data());
        return out;
    }
    // LPCSolver usage (spectrum.rs:40-42): autocorrelate(p+1) [-> normalize] -> lpc(p), one pass over the samples
    static void solve(Context &c, const Frames &f, size_t n_coeffs, bool normalize, double *r_out, double *lpc_out) {
        c.check(vbx_autocorr_lpc_f64(c.get(), f.x, f.n_frames, f.frame_len, f.stride, f.window, n_coeffs, normalize ? 1 : 0, r_out, lpc_out));
    }
    static void lpc_praat_mut(Context &c, const Frames &f, size_t n_coeffs, double *coeffs, int32_t *status) {
        c.check(vbx_lpc_burg_f64(c.get(), f.x, f.n_frames, f.frame_len, f.stride, f.window, n_coeffs, coeffs, status));
    }
    static DeviceVec<double> lpc_praat(Context &c, const Frames &f, size_t n_coeffs, int32_t *status = nullptr) {
        DeviceVec<double> out(c, f.n_frames * n_coeffs);
        lpc_praat_mut(c, f, n_coeffs, out.data(), status);
        return out;
    }
};

struct Pitched {         // periodic.rs:356-358; local_peak / global_peak are unused by the reference and omitted
    static void pitch(Context &c, const Frames &f, double sample_rate, double threshold, double min, double max,
                      size_t kmax, Pitch *candidates /* [F, kmax] */, int32_t *count, int32_t *status) {
        c.check(vbx_pitch_f64(c.get(), f.x, f.n_frames, f.frame_len, f.stride, f.window, sample_rate, threshold, min, max,
                              kmax, candidates, count, status));
    }
};

struct Polynomial {      // polynomial.rs:10-21
    static void find_roots_mut(Context &c, Complex *polys, size_t n_polys, size_t len, int32_t *status) {
        c.check(vbx_find_roots_c64(c.get(), polys, n_polys, len, status));
    }
    static void laguerre(Context &c, const Complex *polys, size_t n_polys, size_t len, Complex start, Complex *out) {
        c.check(vbx_laguerre_c64(c.get(), polys, n_polys, len, start, out));
    }
    static size_t find_roots_work_size(size_t len) { return len * 6 + 4; }   // polynomial.rs:75-77 (unused: the library owns its scratch)
    // self / (x + other): quotient left in polys, remainder in rem (polynomial.rs:155-195)
    static void div_polynomial_mut(Context &c, Complex *polys, const Complex *others, size_t n_polys, size_t len,
                                   Complex *rem, int32_t *status) {
        c.check(vbx_div_polynomial_c64(c.get(), polys, others, n_polys, len, rem, status));
    }
    static size_t degree(const Complex *h_poly, size_t len) { return vbx_degree_c64(h_poly, len); }     // polynomial.rs:26
    static size_t off_low(const Complex *h_poly, size_t len) { return vbx_off_low_c64(h_poly, len); }   // polynomial.rs:30
    // the Complex<f32> instantiation (polynomial.rs:336-386)
    static void find_roots_mut(Context &c, vbx_complex32 *polys, size_t n_polys, size_t len, int32_t *status) {
        c.check(vbx_find_roots_c32(c.get(), polys, n_polys, len, status));
    }
    static void laguerre(Context &c, const vbx_complex32 *polys, size_t n_polys, size_t len, vbx_complex32 start,
                         vbx_complex32 *out) {
        c.check(vbx_laguerre_c32(c.get(), polys, n_polys, len, start, out));
    }
};

struct ToResonance {     // spectrum.rs:195-210
    static void to_resonance(Context &c, const Complex *roots, size_t n_rows, size_t n_roots, double sample_rate,
                             Resonance *out, int32_t *count) {
        c.check(vbx_to_resonance_c64(c.get(), roots, n_rows, n_roots, sample_rate, out, count));
    }
};

// utterance boundaries for the one stateful step of the path
struct Segments {
    const int64_t *h_seg_start = nullptr;   // host, ascending, [0] == 0
    size_t n = 0;
};

struct EstimateFormants {   // spectrum.rs:216-219, iterated as FormantExtractor (:357-369)
    static void estimate_formants(Context &c, const Resonance *res, size_t n_frames, size_t n_res, Segments seg,
                                  const std::vector<Resonance> &starting_estimates, const int32_t *frame_status,
                                  Resonance *out /* [F, n_est] */) {
        c.check(vbx_estimate_formants_f64(c.get(), res, n_frames, n_res, seg.h_seg_start, seg.n, starting_estimates.data(),
                                          starting_estimates.size(), frame_status, out));
    }
};

struct MFCC {            // spectrum.rs:371-373
    static void mfcc(Context &c, const Frames &f, size_t num_coeffs, std::pair<double, double> freq_bounds,
                     double sample_rate, double *out, int32_t *status) {
        c.check(vbx_mfcc_f64(c.get(), f.x, f.n_frames, f.frame_len, f.stride, f.window, num_coeffs, freq_bounds.first,
                             freq_bounds.second, sample_rate, out, status));
    }
    static double hz_to_mel(double hz) { return vbx_hz_to_mel(hz); }     // spectrum.rs:375
    static double mel_to_hz(double mel) { return vbx_mel_to_hz(mel); }   // spectrum.rs:379
    static void dct_mut(Context &c, const double *signal, size_t n_rows, size_t n, double *coeffs) {
        c.check(vbx_dct_f64(c.get(), signal, n_rows, n, coeffs));
    }
};

inline size_t find_formants_real_work_size(size_t buf_len, size_t n_coeffs) { return vbx_find_formants_real_work_size(buf_len, n_coeffs); }
inline size_t find_formants_complex_work_size(size_t n_coeffs) { return vbx_find_formants_complex_work_size(n_coeffs); }

// vox_box::find_formants (lib.rs:40) over a batch.  `formants` receives the
// estimates after every frame ([F, n_est]); frames whose status != Ok leave the state untouched.
inline void find_formants(Context &c, const Frames &f, double sample_rate, size_t n_coeffs, Segments seg,
                          const std::vector<Resonance> &starting_estimates, Resonance *formants,
                          Resonance *resonances = nullptr, int32_t *res_count = nullptr, double *lpc_coeffs = nullptr,
                          int32_t *status = nullptr, double resample_ratio = 1.0) {
    if (f.window != nullptr) throw Error(VBX_E_INVALID, "find_formants applies its own periodic Hanning (lib.rs:65-70): pass rectangular frames");
    if (resample_ratio != 1.0) {   // lib.rs:57-61 (sample-crate arithmetic: parity unpinned), then the same chain on the dense batch
        const size_t m = vbx_resampled_len(f.frame_len, resample_ratio);
        DeviceVec<double> dense(c, f.n_frames * m);
        c.check(vbx_resample_linear_f64(c.get(), f.x, f.n_frames, f.frame_len, f.stride, resample_ratio, dense.data()));
        c.check(vbx_find_formants_f64(c.get(), dense.data(), f.n_frames, m, m, sample_rate, n_coeffs, seg.h_seg_start,
                                      seg.n, starting_estimates.data(), starting_estimates.size(), formants, resonances,
                                      res_count, lpc_coeffs, status));
        c.sync();
        return;
    }
    c.check(vbx_find_formants_f64(c.get(), f.x, f.n_frames, f.frame_len, f.stride, sample_rate, n_coeffs, seg.h_seg_start,
                                  seg.n, starting_estimates.data(), starting_estimates.size(), formants, resonances,
                                  res_count, lpc_coeffs, status));
}

// The user's whole frame loop in one call (examples/pitch_detection.rs:23-30, tests/lib.rs:71-83): per frame
// pitch(..)[0], autocorrelate(p + 1) -> lpc(p), find_formants(..) with the state carried per utterance, mfcc(..), written as
// one record of record_doubles(params) doubles per frame.  `records` [F, record_ld], `status3` [3, F] (optional).
using AnalysisParams = vbx_analysis_params;
inline AnalysisParams analysis_params(double sample_rate, size_t lpc_order = 12, size_t formant_order = 12, size_t mfcc_coeffs = 13) {
    AnalysisParams p{};
    p.sample_rate = sample_rate; p.pitch_threshold = 0.2; p.pitch_fmin = 75.0; p.pitch_fmax = 600.0;
    p.lpc_order = lpc_order; p.formant_order = formant_order; p.n_est = formant_order ? 4 : 0;
    for (size_t e = 0; e < 4; e++) { p.est_init[e].frequency = VBX_MALE_FORMANT_ESTIMATES[e]; p.est_init[e].bandwidth = 1.0; }
    p.mfcc_coeffs = mfcc_coeffs; p.mfcc_lo_hz = 100.0; p.mfcc_hi_hz = 8000.0;
    return p;
}
inline size_t record_doubles(const AnalysisParams &p) { return vbx_record_doubles(&p); }
inline void analyze_frames(Context &c, const Frames &f, const AnalysisParams &p, Segments seg, double *records, size_t record_ld,
                           int32_t *status3 = nullptr) {
    if (f.window != nullptr) throw Error(VBX_E_INVALID, "analyze_frames applies the windows itself: pass rectangular frames");
    c.check(vbx_analyze_frames_f64(c.get(), f.x, f.n_frames, f.frame_len, f.stride, &p, seg.h_seg_start, seg.n, records, record_ld, status3));
}
// the same on the WAV reader's 16-bit PCM (tests/lib.rs:17-19 before the `/ 32767`): bit-identical records, a quarter of the bytes
inline void analyze_frames_pcm16(Context &c, const int16_t *pcm, size_t n_frames, size_t frame_len, size_t stride, const AnalysisParams &p,
                                 Segments seg, double *records, size_t record_ld, int32_t *status3 = nullptr) {
    c.check(vbx_analyze_frames_pcm16(c.get(), pcm, n_frames, frame_len, stride, &p, seg.h_seg_start, seg.n, records, record_ld, status3));
}

// Frame-range sharding of one recording over the GPUs of a node (no counterpart in the reference) and the gather of the
// per-frame records to one rank: grouped ncclSend / ncclRecv inside the library, one communicator per process.
struct Shard { size_t lo, hi, s0, s1; };
inline Shard shard(size_t n_frames, int world, int rank, Segments seg, size_t frame_len, size_t hop) {
    Shard r{};
    if (vbx_shard_range(n_frames, world, rank, seg.h_seg_start, seg.n, &r.lo, &r.hi) != VBX_SUCCESS ||
        vbx_shard_samples(r.lo, r.hi, frame_len, hop, &r.s0, &r.s1) != VBX_SUCCESS)
        throw Error(VBX_E_INVALID, "shard: bad argument");
    return r;
}
// One rank's part of a recording whose utterances the frame split may cut (vbx_shard_plan): its frames, the tracker's warm-up
// frames before them, the utterance starts re-based to the shard, and whether the track continues from / into a neighbour.
struct ShardPlan {
    vbx_shard_plan_t plan{};
    std::vector<int64_t> local_seg_start;
    size_t first() const { return plan.lo - plan.warm; }              // first frame the rank analyses
    size_t n_frames() const { return plan.hi - plan.lo + plan.warm; }  // frames it analyses
};
inline ShardPlan shard_plan(size_t n_frames, int world, int rank, Segments seg) {
    ShardPlan p;
    size_t n = 0;
    if (vbx_shard_plan(n_frames, world, rank, seg.h_seg_start, seg.n, &p.plan) != VBX_SUCCESS ||
        vbx_shard_local_segments(&p.plan, seg.h_seg_start, seg.n, nullptr, 0, &n) != VBX_SUCCESS)
        throw Error(VBX_E_INVALID, "shard_plan: bad argument");
    p.local_seg_start.resize(n);
    if (vbx_shard_local_segments(&p.plan, seg.h_seg_start, seg.n, p.local_seg_start.data(), n, &n) != VBX_SUCCESS)
        throw Error(VBX_E_INVALID, "shard_plan: bad argument");
    return p;
}
// the one-device form of Comm::stitch_tracks (vbx_track_stitch_f64)
inline void track_stitch(Context &c, vbx_resonance *formants, size_t n_frames, size_t formants_ld, size_t first, size_t stop,
                         const vbx_resonance *d_state_in, int32_t *d_changed = nullptr) {
    c.check(vbx_track_stitch_f64(c.get(), formants, n_frames, formants_ld, first, stop, d_state_in, d_changed));
}

class Comm {
public:
    static std::vector<unsigned char> unique_id() {
        std::vector<unsigned char> id(VBX_UNIQUE_ID_BYTES);
        if (vbx_comm_unique_id(id.data()) != VBX_SUCCESS) throw Error(VBX_E_RUNTIME, vbx_last_error(nullptr));
        return id;
    }
    Comm(Context &c, const std::vector<unsigned char> &id, int world, int rank) : ctx_(c), world_(world), rank_(rank) {
        if (id.size() != VBX_UNIQUE_ID_BYTES) throw Error(VBX_E_INVALID, "Comm: the id must hold VBX_UNIQUE_ID_BYTES bytes");
        c.check(vbx_comm_create(c.get(), id.data(), world, rank, &h_));
    }
    ~Comm() { vbx_comm_destroy(h_); }
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    // local: rows[rank] rows of row_doubles doubles; out (dst only): out_doubles doubles, at least what the gather writes
    // (checked here against its transfer list, vbx_gather_plan: the library cannot know the size of a raw device pointer)
    void gather_records(const double *local, const std::vector<int64_t> &rows, size_t row_doubles, int dst, double *out,
                        size_t out_doubles, int slot = 0) {
        if ((int)rows.size() != world_ || dst < 0 || dst >= world_) throw Error(VBX_E_INVALID, "gather_records: one row count per rank, dst a rank");
        std::vector<int64_t> off(rows.size()), cnt(rows.size());
        ctx_.check(vbx_gather_plan(rows.data(), world_, rank_, dst, row_doubles, off.data(), cnt.data(), nullptr));
        if (rank_ == dst && (!out || out_doubles < (size_t)(off.back() + cnt.back())))
            throw Error(VBX_E_INVALID, "gather_records: the gathered buffer is smaller than the rows the ranks send");
        ctx_.check(vbx_gather_records_f64(ctx_.get(), h_, local, rows.data(), row_doubles, dst, out, slot));
    }
    // formants: the formant column of the records the LAST analyze / find_formants call wrote for the plan's frames
    void stitch_tracks(vbx_resonance *formants, size_t formants_ld, const ShardPlan &p, int32_t *d_changed = nullptr, int slot = 0) {
        ctx_.check(vbx_comm_stitch_tracks_f64(ctx_.get(), h_, formants, p.n_frames(), formants_ld, &p.plan, d_changed, slot));
    }
    void wait(int slot) { ctx_.check(vbx_comm_wait(ctx_.get(), h_, slot)); }
    void sync() { ctx_.check(vbx_comm_sync(h_)); }
private:
    Context &ctx_;
    vbx_comm *h_ = nullptr;
    int world_ = 1, rank_ = 0;
};

}  // namespace voxbox
