// vbx_api.hip -- host side of libvoxbox_hip.so: context, workspaces, host-built tables and
// the extern "C" entry points declared in include/voxbox_hip.h.  No CPU fallback: every
// numeric entry point launches gfx950 kernels on the context's stream.
#include "../../include/voxbox_hip.h"
#include "vbx_kernels.hpp"
#include "vbx_host.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <array>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <string>
#include <tuple>
#include <vector>

using namespace vbx;

extern "C" const double VBX_MALE_FORMANT_ESTIMATES[4] = {320., 1440., 2760., 3200.};     // src/lib.rs:27
extern "C" const double VBX_FEMALE_FORMANT_ESTIMATES[4] = {480., 1760., 3200., 3520.};   // src/lib.rs:28

namespace {

thread_local std::string g_last_error;

struct ProfRec { std::string name; hipEvent_t a, b; hipStream_t stream; };

}  // namespace

struct vbx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string last_error;
    size_t curve_failed_bytes = 0;                        // a WS_CURVE size hipMalloc refused (launch_spectral: not retried per call)
    std::string arch;
    int cu_count = 0;
    // workspaces (grown on demand, never shrunk)
    enum { WS_COEFFS, WS_RES, WS_COUNT, WS_STATUS, WS_MISC, WS_SEG, WS_EST, WS_UNSURE, WS_F32_IN, WS_F32_OUT, WS_TRK, WS_BURG_LIST, WS_ROOTS_LIST, WS_LONG, WS_LONG2, WS_CZT, WS_CURVE, WS_LPC_LIST, WS_N };
    void *ws[WS_N] = {nullptr};
    const int32_t *burg_list_count = nullptr;             // device counter of the last one-pass Burg call (tests)
    const int32_t *roots_list_count = nullptr;            // the same for the resonance kernel of find_formants
    size_t ws_bytes[WS_N] = {0};
    // cached device tables
    std::map<std::pair<int, size_t>, double *> windows;   // (kind, n)
    std::map<size_t, bool> lag_rcp_ok;                    // n -> the lag window's table carries usable reciprocals (get_window_dev)
    std::map<size_t, float *> lag_windows32;              // n -> the lag window table rounded to f32 (Pitched<f32, f32>)
    std::map<std::tuple<size_t, int, int>, double *> goertzel;   // (n, b_lo, nb) -> [nb][2] kappa, sigma
    std::map<size_t, double *> dct_tables;                // K -> [K][K]
    std::map<std::pair<size_t, int>, std::pair<double *, double *>> dft2_tabs;   // (n, n1) -> (stage-1 table, twiddles)
    std::map<std::tuple<size_t, int, int>, std::array<double *, 4>> mfma_tabs;   // (n, n1, k2) -> ctab, twd, twm, wm
    std::map<std::tuple<size_t, int, int, int>, std::pair<double *, double *>> czt_tabs;   // (n, top, L, split length or 0) -> (chirp, FFT of the chirp segment(s))
    bool pitch_whole_curve = false;                       // VBX_PITCH_CURVE_CUT=0: the pow2 kernels keep every lag of the curve in LDS (tests)
    bool mfcc_czt_split = false;                          // VBX_MFCC_CZT_SPLIT=1: the two-block form of the chirp-z kernel wherever it fits (tests)
    int mfcc_czt = -1;                                    // VBX_MFCC_CZT=0 / 1: never / wherever it fits (tests); -1: the measured choice
    std::map<std::tuple<size_t, size_t, double, double, double>, int32_t *> bins_cache;
    std::map<std::tuple<size_t, size_t, double, double, double>, double *> slopes_cache;   // [nb][2] i/up, i/down per bin
    std::map<std::tuple<int, int, int, int>, std::pair<void *, mfcc_interp_t>> interp_cache;   // (plan, n, b_lo, nb) -> tables of the interpolated MFCC bins (first == nullptr: no such form)
    int last_mfcc_interp = 0;                             // the last vbx_mfcc_f64 call took the interpolated form (tests)
    int last_spectral_split = 0;                          // the last fused / pitch call ran as two kernels (tests)
    int mfcc_defer = 1;                                   // VBX_MFCC_DEFER=0: log10 + DCT of the fused call's MFCC rows inside the frame's wavefront (rounds 2-5; A/B)
    int lpc_exact = 1;                                    // VBX_LPC_EXACT=0: no conditioning probe, no double-double redo of flagged LPC rows (rounds 1-5; tests, A/B)
    int pow2_split = -1;                                  // VBX_POW2_SPLIT=0: the 4096-point plan as ONE kernel (transforms and refinement fused, as before round 5; tests, A/B)
    int mfcc_interp = -1;                                 // VBX_MFCC_INTERP=0: never (the chirp-z kernel beside the fused one, as before round 5; tests, A/B)
    std::map<std::pair<size_t, double>, std::pair<int32_t *, double *>> resample_tabs;   // (n, ratio) -> (index, fraction)
    // timing
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool prof = false;
    // VBX_ROCTX=1: every kernel group of an entry point (the names vbx_profile_* reports) is also a roctx range, so that a
    // `rocprofv3 --marker-trace --kernel-trace` timeline shows which call a kernel belongs to
    int (*roctx_push)(const char *) = nullptr;
    int (*roctx_pop)() = nullptr;
    std::vector<ProfRec> recs;
    std::map<std::string, std::pair<double, long>> prof_acc;
    std::map<std::string, int> prof_stream;               // name -> 0: the context's stream, 1: the side stream, 2: the tracker's
    double *spectral_tab[SPECTRAL_PLANS] = {};             // twiddles of k_spectral*.hip, by plan
    bool pitch_force_mfma = false;                        // test hook: VBX_PITCH_MFMA=1 keeps the matrix-core pitch kernel on 1200-sample frames
    bool mfcc_force_goertzel = false;                     // test hooks: VBX_MFCC_GOERTZEL=1 / VBX_MFCC_DFT2=1 keep the
    bool mfcc_force_dft2 = false;                         //   fallback kernels covered on lengths the MFMA kernel takes
    bool mfcc_force_mfma = false;                         //   VBX_MFCC_MFMA=1: no FFT kernel for 1024 / 1200 / 2048 / 4096 (k_spectral*.hip)
    unsigned long long *pitch_work = nullptr;             // [PITCH_WORK_SLOTS][4], counted while profiling
    // second stream of vbx_analyze_frames_f64 (the formant chain runs beside the pitch kernel) + fork/join events
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t trk = nullptr;                            // the tracker's time slices (run_find_formants)
    hipEvent_t ev_slice[8] = {nullptr}, ev_trk = nullptr;
    // pinned staging of the small host arrays (segment starts, initial estimates): the caller's arrays may be
    // freed on return, and an upload whose content has not changed since the last call is skipped
    void *stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    std::vector<char> staged[2];                          // content now on the device
    hipEvent_t stage_ev[2] = {nullptr, nullptr};          // completion of the last staged copy
    // what the tracker of the last find_formants / analyze_frames call ran on (vbx_track_stitch_f64 continues that track)
    struct { const res_t *res = nullptr; const int32_t *cnt = nullptr, *st = nullptr; long F = 0; int n_est = 0;
             res_t *out = nullptr; long out_ld = 0; } last_track;
    double *stitch_state = nullptr;                       // 2 * VBX_FORMANT_SLOTS doubles: the state a stitch received (vbx_comm.hip)
};

namespace {

int fail(vbx_ctx *ctx, int code, const std::string &msg) {
    g_last_error = msg;
    if (ctx) ctx->last_error = msg;
    return code;
}

#define VBX_HIP(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, VBX_E_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define VBX_REQUIRE(ctx, cond, msg) \
    do { if (!(cond)) return fail(ctx, VBX_E_INVALID, std::string(__func__) + ": " + (msg)); } while (0)

int check_launch(vbx_ctx *ctx, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, VBX_E_RUNTIME, std::string(what) + " launch: " + hipGetErrorString(e));
    return VBX_SUCCESS;
}

struct Prof {
    vbx_ctx *ctx; const char *name; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    Prof(vbx_ctx *c, const char *n, hipStream_t stream = nullptr) : ctx(c), name(n), st(stream ? stream : c->stream) {
        if (ctx->roctx_push) ctx->roctx_push(name);
        if (ctx->prof) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, st); }
    }
    ~Prof() {
        if (ctx->prof) { hipEventRecord(b, st); ctx->recs.push_back({name, a, b, st}); }
        if (ctx->roctx_pop) ctx->roctx_pop();
    }
};

int ws_get(vbx_ctx *ctx, int slot, size_t bytes, void **out) {
    if (bytes == 0) bytes = 16;
    if (ctx->ws_bytes[slot] < bytes) {
        if (ctx->ws[slot]) {
            VBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->side) VBX_HIP(ctx, hipStreamSynchronize(ctx->side));
            if (ctx->trk) VBX_HIP(ctx, hipStreamSynchronize(ctx->trk));
            VBX_HIP(ctx, hipFree(ctx->ws[slot]));
            ctx->ws[slot] = nullptr; ctx->ws_bytes[slot] = 0;
        }
        size_t cap = bytes + bytes / 8;
        VBX_HIP(ctx, hipMalloc(&ctx->ws[slot], cap));
        ctx->ws_bytes[slot] = cap;
    }
    *out = ctx->ws[slot];
    return VBX_SUCCESS;
}

// ---- host-built tables (the library's own statement of the sample-crate recurrences) --------

// The lag window's table carries its entries' reciprocals behind them (each correctly rounded: the host's IEEE division), from
// element (n + 1) & ~1 on, for the fused kernels' lag-window divide (quotient_by_table, vbx_spectral.hpp); ctx->lag_rcp_ok[n] says
// whether they may be used (no zero, no entry whose reciprocal leaves the normal range).
int get_window_dev(vbx_ctx *ctx, int kind, size_t n, const double **out) {
    auto key = std::make_pair(kind, n);
    auto it = ctx->windows.find(key);
    if (it == ctx->windows.end()) {
        const size_t off = (n + 1) & ~(size_t)1, total = (kind == VBX_WINDOW_HANNING_LAG) ? off + n : n;
        std::vector<double> h(total, 0.0);
        if (window_table_host(kind, n, h.data()) != VBX_SUCCESS) return fail(ctx, VBX_E_INVALID, "bad window kind");
        if (kind == VBX_WINDOW_HANNING_LAG) {
            bool usable = true;
            for (size_t i = 0; i < n; i++) {
                h[off + i] = 1.0 / h[i];
                usable = usable && std::isfinite(h[off + i]) && std::fabs(h[off + i]) < 1e290 && std::fabs(h[off + i]) > 1e-290;
            }
            ctx->lag_rcp_ok[n] = usable;
        }
        double *d = nullptr;
        VBX_HIP(ctx, hipMalloc((void **)&d, total * sizeof(double)));
        VBX_HIP(ctx, hipMemcpy(d, h.data(), total * sizeof(double), hipMemcpyHostToDevice));
        it = ctx->windows.emplace(key, d).first;
    }
    *out = it->second;
    return VBX_SUCCESS;
}

// Goertzel-Reinsch constants of bins [b_lo, b_lo + nb) of an n-point DFT (k_mfcc.hip)
int get_goertzel_dev(vbx_ctx *ctx, size_t n, int b_lo, int nb, const double **out) {
    auto key = std::make_tuple(n, b_lo, nb);
    auto it = ctx->goertzel.find(key);
    if (it == ctx->goertzel.end()) {
        std::vector<double> h(2 * (size_t)(nb > 0 ? nb : 1));
        for (int i = 0; i < nb; i++) {
            const double w = 2.0 * M_PI * (double)((size_t)(b_lo + i) % n) / (double)n;
            if (std::cos(w) > 0.0) { const double sh = std::sin(0.5 * w); h[2 * i] = 4.0 * sh * sh; h[2 * i + 1] = 1.0; }
            else { const double ch = std::cos(0.5 * w); h[2 * i] = 4.0 * ch * ch; h[2 * i + 1] = -1.0; }
        }
        double *d = nullptr;
        VBX_HIP(ctx, hipMalloc((void **)&d, h.size() * sizeof(double)));
        VBX_HIP(ctx, hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
        it = ctx->goertzel.emplace(key, d).first;
    }
    *out = it->second;
    return VBX_SUCCESS;
}

// tables of the two-stage MFCC DFT (k_mfcc.hip): ctab[i1][c] = cos / sin columns of the n1-point DFT,
// twid[j] = (cos, sin)(2 pi j / n); evaluated in long double and rounded once
int get_dft2_dev(vbx_ctx *ctx, size_t n, const mfcc_plan_t &pl, const double **ctab, const double **twid) {
    auto key = std::make_pair(n, pl.n1);
    auto it = ctx->dft2_tabs.find(key);
    if (it == ctx->dft2_tabs.end()) {
        const long double two_pi = 6.283185307179586476925286766559005768L;
        const int n1 = pl.n1, nc = pl.nc, ncos = n1 / 2 + 1;
        std::vector<double> hc((size_t)n1 * nc, 0.0), ht(2 * n);
        for (int i1 = 0; i1 < n1; i1++)
            for (int c = 0; c < n1; c++) {
                const int k1 = (c < ncos) ? c : c - ncos + 1;
                const long double ang = two_pi * (long double)((long)i1 * k1 % n1) / (long double)n1;
                hc[(size_t)i1 * nc + c] = (double)((c < ncos) ? cosl(ang) : sinl(ang));
            }
        for (size_t j = 0; j < n; j++) {
            const long double ang = two_pi * (long double)j / (long double)n;
            ht[2 * j] = (double)cosl(ang); ht[2 * j + 1] = (double)sinl(ang);
        }
        double *dc = nullptr, *dt = nullptr;
        VBX_HIP(ctx, hipMalloc((void **)&dc, hc.size() * sizeof(double)));
        VBX_HIP(ctx, hipMalloc((void **)&dt, ht.size() * sizeof(double)));
        VBX_HIP(ctx, hipMemcpy(dc, hc.data(), hc.size() * sizeof(double), hipMemcpyHostToDevice));
        VBX_HIP(ctx, hipMemcpy(dt, ht.data(), ht.size() * sizeof(double), hipMemcpyHostToDevice));
        it = ctx->dft2_tabs.emplace(key, std::make_pair(dc, dt)).first;
    }
    *ctab = it->second.first; *twid = it->second.second;
    return VBX_SUCCESS;
}

// tables of the matrix-core MFCC kernel (k_mfcc_mfma.hip), evaluated in long double and rounded once
int get_mfcc_mfma_dev(vbx_ctx *ctx, size_t n, const mfcc_mplan_t &pl, const double **ctab, const double **twd,
                      const double **twm, const double **wm) {
    auto key = std::make_tuple(n, pl.n1, pl.k2);
    auto it = ctx->mfma_tabs.find(key);
    if (it == ctx->mfma_tabs.end()) {
        const long double two_pi = 6.283185307179586476925286766559005768L;
        const int n1 = pl.n1, n2 = pl.n2, n1p = (n1 + 3) & ~3, nc = 32 * pl.ntd, mt = pl.mt;
        std::vector<double> hc((size_t)n1p * nc, 0.0);
        for (int i1 = 0; i1 < n1; i1++)
            for (int c = 0; c < nc; c++) {
                const bool is_sin = c >= 16 * pl.ntd;
                const int k1 = is_sin ? c - 16 * pl.ntd : c;
                if (k1 >= n1) continue;
                const long double ang = two_pi * (long double)((long)i1 * k1 % n1) / (long double)n1;
                hc[(size_t)i1 * nc + c] = (double)(is_sin ? sinl(ang) : cosl(ang));
            }
        auto twiddle = [&](int i2, int k1, double *dst) {
            if (i2 >= n2 || k1 < 0 || k1 >= n1) { dst[0] = 0.0; dst[1] = 0.0; return; }
            const long double ang = two_pi * (long double)((long)i2 * k1 % (long)n) / (long double)n;
            dst[0] = (double)cosl(ang); dst[1] = (double)sinl(ang);
        };
        std::vector<double> hd((size_t)mt * pl.ntd * 4 * 128 + 2, 0.0), hm((size_t)mt * pl.ntm * 4 * 128 + 2, 0.0);
        for (int m = 0; m < mt; m++)
            for (int r = 0; r < 4; r++)
                for (int l = 0; l < 64; l++) {
                    const int i2 = 16 * m + 4 * r + (l >> 4), col = l & 15;
                    for (int t = 0; t < pl.ntd; t++)
                        twiddle(i2, 16 * t + col, &hd[((size_t)((m * pl.ntd + t) * 4 + r) * 64 + l) * 2]);
                    for (int t = 0; t < pl.ntm; t++) {
                        const int kp = 16 * (t == 0 ? pl.src0 : pl.src1) + col;
                        twiddle(i2, (kp >= 1) ? n1 - kp : -1, &hm[((size_t)((m * pl.ntm + t) * 4 + r) * 64 + l) * 2]);
                    }
                }
        // stage-2 A operand: Wm[2 k2 + p][kk], kk = Re rows i2 (0 .. 16 mt) then Im rows; lane l of K-step s holds
        // Wm[l & 15][4 s + (l >> 4)]
        std::vector<double> hw((size_t)8 * mt * 64, 0.0);
        for (int s = 0; s < 8 * mt; s++)
            for (int l = 0; l < 64; l++) {
                const int rowm = l & 15, kk = 4 * s + (l >> 4);
                const bool im_half = kk >= 16 * mt;
                const int i2 = im_half ? kk - 16 * mt : kk, k2 = rowm >> 1, p = rowm & 1;
                if (i2 >= n2 || k2 >= pl.k2) continue;
                const long double ang = two_pi * (long double)((long)i2 * k2 % n2) / (long double)n2;
                const long double c = cosl(ang), sn = sinl(ang);
                // (Bre + i Bim)(c - i sn): Re = Bre c + Bim sn, Im = Bim c - Bre sn
                hw[(size_t)s * 64 + l] = (double)(p == 0 ? (im_half ? sn : c) : (im_half ? c : -sn));
            }
        std::array<double *, 4> d{nullptr, nullptr, nullptr, nullptr};
        const std::vector<double> *src[4] = {&hc, &hd, &hm, &hw};
        for (int i = 0; i < 4; i++) {
            VBX_HIP(ctx, hipMalloc((void **)&d[i], src[i]->size() * sizeof(double)));
            VBX_HIP(ctx, hipMemcpy(d[i], src[i]->data(), src[i]->size() * sizeof(double), hipMemcpyHostToDevice));
        }
        it = ctx->mfma_tabs.emplace(key, d).first;
    }
    *ctab = it->second[0]; *twd = it->second[1]; *twm = it->second[2]; *wm = it->second[3];
    return VBX_SUCCESS;
}

// tables of the chirp-z MFCC kernel (k_mfcc_czt.hip)
int get_czt_dev(vbx_ctx *ctx, size_t n, int top, int L, int n1, const double **chirp, const double **bhat) {
    auto key = std::make_tuple(n, top, L, n1);
    auto it = ctx->czt_tabs.find(key);
    if (it == ctx->czt_tabs.end()) {
        const size_t nblk = (n1 > 0 && (size_t)n1 < n) ? (n + n1 - 1) / n1 : 1;
        std::vector<double> hc(2 * n), hb(2 * (size_t)L * nblk);
        mfcc_czt_fill_tabs((int)n, top, L, n1, hc.data(), hb.data());
        double *dc = nullptr, *db = nullptr;
        VBX_HIP(ctx, hipMalloc((void **)&dc, hc.size() * sizeof(double)));
        VBX_HIP(ctx, hipMalloc((void **)&db, hb.size() * sizeof(double)));
        VBX_HIP(ctx, hipMemcpy(dc, hc.data(), hc.size() * sizeof(double), hipMemcpyHostToDevice));
        VBX_HIP(ctx, hipMemcpy(db, hb.data(), hb.size() * sizeof(double), hipMemcpyHostToDevice));
        it = ctx->czt_tabs.emplace(key, std::make_pair(dc, db)).first;
    }
    *chirp = it->second.first; *bhat = it->second.second;
    return VBX_SUCCESS;
}

int get_dct_dev(vbx_ctx *ctx, size_t k, const double **out) {
    auto it = ctx->dct_tables.find(k);
    if (it == ctx->dct_tables.end()) {
        std::vector<double> h(k * k);
        for (size_t kk = 0; kk < k; kk++)          // src/spectrum.rs:395
            for (size_t n = 0; n < k; n++)
                h[kk * k + n] = std::cos(M_PI * (double)kk * (2. * (double)n + 1.) / (2. * (double)k));
        double *d = nullptr;
        VBX_HIP(ctx, hipMalloc((void **)&d, k * k * sizeof(double)));
        VBX_HIP(ctx, hipMemcpy(d, h.data(), k * k * sizeof(double), hipMemcpyHostToDevice));
        it = ctx->dct_tables.emplace(k, d).first;
    }
    *out = it->second;
    return VBX_SUCCESS;
}

// tables of the MFCC bins interpolated inside the fused kernel (mfcc_interp_t); *ok = false: the shape has no such form
int get_interp_dev(vbx_ctx *ctx, int plan, int n, int b_lo, int nb, mfcc_interp_t *out, bool *ok) {
    auto key = std::make_tuple(plan, n, b_lo, nb);
    auto it = ctx->interp_cache.find(key);
    if (it == ctx->interp_cache.end()) {
        // a caller that sweeps frame lengths or bands would grow the cache without bound (48-200 KB of device memory per entry):
        // past 64 entries everything is dropped (every stream drained first: a queued kernel may still read a table)
        if (ctx->interp_cache.size() >= 64) {
            VBX_HIP(ctx, hipDeviceSynchronize());
            for (auto &kv : ctx->interp_cache) if (kv.second.first) (void)hipFree(kv.second.first);
            ctx->interp_cache.clear();
        }
        mfcc_interp_t d{};
        void *dev = nullptr;
        {
            const size_t bytes = mfcc_interp_table_bytes(plan, nb);
            std::vector<char> h(bytes, 0);
            if (mfcc_interp_fill(plan, n, b_lo, nb, h.data(), &d)) {
                VBX_HIP(ctx, hipMalloc(&dev, bytes));
                if (hipMemcpy(dev, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
                    (void)hipFree(dev);
                    return fail(ctx, VBX_E_RUNTIME, "get_interp_dev: hipMemcpy of the interpolation tables failed");
                }
                char *b = static_cast<char *>(dev);
                d.rot = reinterpret_cast<const double *>(b);
                d.coef = reinterpret_cast<const double *>(b + mfcc_interp_coef_offset(plan));
                d.j0 = reinterpret_cast<const int32_t *>(b + mfcc_interp_j0_offset(plan, nb));
            }
        }
        it = ctx->interp_cache.emplace(key, std::make_pair(dev, d)).first;
    }
    *ok = it->second.first != nullptr;
    *out = it->second.second;
    return VBX_SUCCESS;
}

// twiddles of the fused spectral kernels (k_spectral.hip, k_spectral_pow2.hip), one table per plan
int get_spectral_tab(vbx_ctx *ctx, int plan, const double **out) {
    if (!ctx->spectral_tab[plan]) {
        std::vector<double> h(2 * (size_t)spectral_tab_complex(plan));
        spectral_fill_tab(plan, h.data());
        VBX_HIP(ctx, hipMalloc((void **)&ctx->spectral_tab[plan], h.size() * sizeof(double)));
        VBX_HIP(ctx, hipMemcpy(ctx->spectral_tab[plan], h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    *out = ctx->spectral_tab[plan];
    return VBX_SUCCESS;
}

int get_bins_dev(vbx_ctx *ctx, size_t n, size_t k, double lo, double hi, double sr,
                 const int32_t **out, std::vector<int32_t> &host_bins, bool &bad) {
    mel_bins_host(n, k, lo, hi, sr, host_bins, bad);
    for (size_t i = 0; i + 1 < host_bins.size(); i++) if (host_bins[i + 1] < host_bins[i]) bad = true;   // usize underflow panics
    if (host_bins.back() > (int32_t)n) bad = true;                                                    // spectrum[bin] out of bounds
    auto key = std::make_tuple(n, k, lo, hi, sr);
    auto it = ctx->bins_cache.find(key);
    if (it == ctx->bins_cache.end()) {
        int32_t *d = nullptr;
        VBX_HIP(ctx, hipMalloc((void **)&d, host_bins.size() * sizeof(int32_t)));
        VBX_HIP(ctx, hipMemcpy(d, host_bins.data(), host_bins.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        it = ctx->bins_cache.emplace(key, d).first;
    }
    *out = it->second;
    return VBX_SUCCESS;
}

// src/spectrum.rs:424,430: the slope factor of every bin, (i as f64) / (up as f64) on the rising side of its
// filter and i / down on the "falling" side (Q14: it rises too); one IEEE division each, as in the reference
int get_slopes_dev(vbx_ctx *ctx, size_t n, size_t k, double lo, double hi, double sr,
                   const std::vector<int32_t> &hb, const double **out) {
    auto key = std::make_tuple(n, k, lo, hi, sr);
    auto it = ctx->slopes_cache.find(key);
    if (it == ctx->slopes_cache.end()) {
        const int b_lo = hb.front(), nb = hb.back() - hb.front();
        std::vector<double> h(2 * (size_t)(nb > 0 ? nb : 1), 0.0);
        for (size_t w = 0; w < k; w++) {
            const int up = hb[w + 1] - hb[w], down = hb[w + 2] - hb[w + 1];
            for (int i = 0; i < up; i++) h[2 * (size_t)(hb[w] + i - b_lo)] = (double)i / (double)up;
            for (int i = 0; i < down; i++) h[2 * (size_t)(hb[w + 1] + i - b_lo) + 1] = (double)i / (double)down;
        }
        double *d = nullptr;
        VBX_HIP(ctx, hipMalloc((void **)&d, h.size() * sizeof(double)));
        VBX_HIP(ctx, hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
        it = ctx->slopes_cache.emplace(key, d).first;
    }
    *out = it->second;
    return VBX_SUCCESS;
}

// max_len: VBX_MAX_FRAME_LEN for the entry points whose kernels keep a frame in registers / LDS, VBX_MAX_LONG_FRAME_LEN for
// the ones that also have a tiled form for longer frames (k_long.hip)
int check_frames(vbx_ctx *ctx, const char *fn, const void *x, size_t n_frames, size_t frame_len, size_t stride,
                 size_t max_len = VBX_MAX_FRAME_LEN) {
    if (!ctx) return fail(nullptr, VBX_E_INVALID, std::string(fn) + ": null context");
    if (n_frames == 0) return 1;   // empty batch: nothing to do
    if (!x) return fail(ctx, VBX_E_INVALID, std::string(fn) + ": null frame pointer");
    if (frame_len < 1 || frame_len > max_len)
        return fail(ctx, VBX_E_INVALID, std::string(fn) + ": frame_len must be in [1, " + std::to_string(max_len) + "]");
    if (stride < 1) return fail(ctx, VBX_E_INVALID, std::string(fn) + ": stride must be >= 1");
    if (n_frames > 0x7fffffffull) return fail(ctx, VBX_E_INVALID, std::string(fn) + ": too many frames for one launch");
    return VBX_SUCCESS;
}

// Small host array -> ctx-owned device buffer through pinned staging.  The caller's array may be freed on return
// (the copy below reads the staging buffer, not the caller's memory), the host is not blocked behind queued work,
// and identical content (the usual case: the same segments / estimates every call) is not uploaded again.
int stage_upload(vbx_ctx *ctx, int which, int ws_slot, const void *h_src, size_t bytes, hipStream_t st, void **d_out) {
    void *d = nullptr;
    const bool grown = ctx->ws_bytes[ws_slot] < bytes;
    int rc = ws_get(ctx, ws_slot, bytes, &d);
    if (rc != VBX_SUCCESS) return rc;
    *d_out = d;
    std::vector<char> &have = ctx->staged[which];
    if (!grown && have.size() == bytes && std::memcmp(have.data(), h_src, bytes) == 0) return VBX_SUCCESS;
    if (!ctx->stage_ev[which]) VBX_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_ev[which], hipEventDisableTiming));
    else VBX_HIP(ctx, hipEventSynchronize(ctx->stage_ev[which]));          // the previous copy has left the staging buffer
    if (ctx->stage_cap[which] < bytes) {
        if (ctx->stage[which]) VBX_HIP(ctx, hipHostFree(ctx->stage[which]));
        ctx->stage[which] = nullptr; ctx->stage_cap[which] = 0;
        VBX_HIP(ctx, hipHostMalloc(&ctx->stage[which], bytes + bytes / 2, hipHostMallocDefault));
        ctx->stage_cap[which] = bytes + bytes / 2;
    }
    std::memcpy(ctx->stage[which], h_src, bytes);
    // uploads of either stream are ordered behind everything queued on the main stream that may still read the old content
    VBX_HIP(ctx, hipMemcpyAsync(d, ctx->stage[which], bytes, hipMemcpyHostToDevice, st));
    VBX_HIP(ctx, hipEventRecord(ctx->stage_ev[which], st));
    have.assign((const char *)h_src, (const char *)h_src + bytes);
    return VBX_SUCCESS;
}

int upload_segments(vbx_ctx *ctx, hipStream_t st, const int64_t *h_seg, size_t n_seg, size_t n_frames, const int64_t **d_seg, size_t *n_out) {
    if (h_seg == nullptr || n_seg == 0) { *d_seg = nullptr; *n_out = 1; return VBX_SUCCESS; }
    if (h_seg[0] != 0) return fail(ctx, VBX_E_INVALID, "seg_start[0] must be 0");
    for (size_t i = 1; i < n_seg; i++)
        if (h_seg[i] < h_seg[i - 1] || (size_t)h_seg[i] > n_frames) return fail(ctx, VBX_E_INVALID, "seg_start must ascend within [0, n_frames]");
    void *d = nullptr;
    int rc = stage_upload(ctx, 0, vbx_ctx::WS_SEG, h_seg, n_seg * sizeof(int64_t), st, &d);
    if (rc != VBX_SUCCESS) return rc;
    *d_seg = (const int64_t *)d; *n_out = n_seg;
    return VBX_SUCCESS;
}

int upload_estimates(vbx_ctx *ctx, hipStream_t st, const vbx_resonance *h_est, size_t n_est, const res_t **d_est) {
    void *d = nullptr;
    int rc = stage_upload(ctx, 1, vbx_ctx::WS_EST, h_est, n_est * sizeof(vbx_resonance), st, &d);
    if (rc != VBX_SUCCESS) return rc;
    *d_est = (const res_t *)d;
    return VBX_SUCCESS;
}

}  // namespace

// =============================================================================================
extern "C" {

// hooks for vbx_comm.hip (the context is opaque outside this file)
int vbx_internal_fail(vbx_ctx *ctx, int code, const char *msg) { return fail(ctx, code, msg ? msg : ""); }
void *vbx_internal_stream(vbx_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int vbx_internal_device(vbx_ctx *ctx) { return ctx ? ctx->device : 0; }

int vbx_abi_version(void) { return VBX_ABI_VERSION; }

int vbx_ctx_create(vbx_ctx **out, int device, void *hip_stream) {
    if (!out) return fail(nullptr, VBX_E_INVALID, "vbx_ctx_create: null out");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, VBX_E_NODEVICE, "vbx_ctx_create: no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= count) return fail(nullptr, VBX_E_INVALID, "vbx_ctx_create: bad device ordinal");
    VBX_HIP(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    VBX_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, VBX_E_NODEVICE, std::string("vbx_ctx_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    vbx_ctx *ctx = new vbx_ctx();
    ctx->device = device;
    ctx->arch = prop.gcnArchName;
    ctx->cu_count = prop.multiProcessorCount;
    { const char *e = std::getenv("VBX_MFCC_GOERTZEL"); ctx->mfcc_force_goertzel = e && e[0] == '1'; }
    { const char *e = std::getenv("VBX_MFCC_DFT2"); ctx->mfcc_force_dft2 = e && e[0] == '1'; }
    { const char *e = std::getenv("VBX_MFCC_MFMA"); ctx->mfcc_force_mfma = e && e[0] == '1'; }
    { const char *e = std::getenv("VBX_MFCC_CZT"); ctx->mfcc_czt = e ? (e[0] == '1' ? 1 : 0) : -1; }
    { const char *e = std::getenv("VBX_MFCC_DEFER"); ctx->mfcc_defer = (e && e[0] == '0') ? 0 : 1; }
    { const char *e = std::getenv("VBX_LPC_EXACT"); ctx->lpc_exact = (e && e[0] == '0') ? 0 : 1; }
    { const char *e = std::getenv("VBX_POW2_SPLIT"); ctx->pow2_split = e ? (e[0] == '0' ? 0 : 1) : -1; }
    { const char *e = std::getenv("VBX_MFCC_INTERP"); ctx->mfcc_interp = e ? (e[0] == '0' ? 0 : 1) : -1; }
    { const char *e = std::getenv("VBX_MFCC_CZT_SPLIT"); ctx->mfcc_czt_split = e != nullptr && e[0] == '1'; }
    { const char *e = std::getenv("VBX_PITCH_CURVE_CUT"); ctx->pitch_whole_curve = e != nullptr && e[0] == '0'; }
    if (const char *e = std::getenv("VBX_ROCTX"); e != nullptr && e[0] == '1') {
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (h == nullptr) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h != nullptr) {
            ctx->roctx_push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
            ctx->roctx_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (ctx->roctx_push == nullptr || ctx->roctx_pop == nullptr) { ctx->roctx_push = nullptr; ctx->roctx_pop = nullptr; }
        }
    }
    { const char *e = std::getenv("VBX_PITCH_MFMA"); ctx->pitch_force_mfma = e && e[0] == '1'; }
    if (hip_stream) { ctx->stream = (hipStream_t)hip_stream; ctx->owns_stream = false; }
    else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return fail(nullptr, VBX_E_RUNTIME, "hipStreamCreate failed"); }
        ctx->owns_stream = true;
    }
    hipEventCreate(&ctx->t0);
    hipEventCreate(&ctx->t1);
    *out = ctx;
    return VBX_SUCCESS;
}

void vbx_ctx_destroy(vbx_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < vbx_ctx::WS_N; i++) if (ctx->ws[i]) hipFree(ctx->ws[i]);
    if (ctx->pitch_work) hipFree(ctx->pitch_work);
    if (ctx->stitch_state) hipFree(ctx->stitch_state);
    for (double *t : ctx->spectral_tab) if (t) hipFree(t);
    for (auto &kv : ctx->windows) hipFree(kv.second);
    for (auto &kv : ctx->lag_windows32) hipFree(kv.second);
    for (auto &kv : ctx->goertzel) hipFree(kv.second);
    for (auto &kv : ctx->dct_tables) hipFree(kv.second);
    for (auto &kv : ctx->dft2_tabs) { hipFree(kv.second.first); hipFree(kv.second.second); }
    for (auto &kv : ctx->mfma_tabs) for (double *q : kv.second) hipFree(q);
    for (auto &kv : ctx->czt_tabs) { hipFree(kv.second.first); hipFree(kv.second.second); }
    for (auto &kv : ctx->bins_cache) hipFree(kv.second);
    for (auto &kv : ctx->slopes_cache) hipFree(kv.second);
    for (auto &kv : ctx->interp_cache) if (kv.second.first) hipFree(kv.second.first);
    for (auto &kv : ctx->resample_tabs) { hipFree(kv.second.first); hipFree(kv.second.second); }
    for (auto &r : ctx->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    if (ctx->side) { hipStreamSynchronize(ctx->side); hipStreamDestroy(ctx->side); }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    for (auto &e : ctx->ev_slice) if (e) hipEventDestroy(e);
    if (ctx->ev_trk) hipEventDestroy(ctx->ev_trk);
    if (ctx->trk) { hipStreamSynchronize(ctx->trk); hipStreamDestroy(ctx->trk); }
    for (int i = 0; i < 2; i++) { if (ctx->stage[i]) hipHostFree(ctx->stage[i]); if (ctx->stage_ev[i]) hipEventDestroy(ctx->stage_ev[i]); }
    if (ctx->t0) hipEventDestroy(ctx->t0);
    if (ctx->t1) hipEventDestroy(ctx->t1);
    if (ctx->owns_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int vbx_sync(vbx_ctx *ctx) {
    if (!ctx) return fail(nullptr, VBX_E_INVALID, "vbx_sync: null context");
    VBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VBX_SUCCESS;
}

const char *vbx_last_error(const vbx_ctx *ctx) { return ctx ? ctx->last_error.c_str() : g_last_error.c_str(); }

int vbx_device_info(const vbx_ctx *ctx, char *h_name, size_t name_cap, int *h_cu_count) {
    if (!ctx) return fail(nullptr, VBX_E_INVALID, "vbx_device_info: null context");
    if (h_name && name_cap) { std::strncpy(h_name, ctx->arch.c_str(), name_cap - 1); h_name[name_cap - 1] = 0; }
    if (h_cu_count) *h_cu_count = ctx->cu_count;
    return VBX_SUCCESS;
}

int vbx_malloc(vbx_ctx *ctx, void **out_dptr, size_t bytes) {
    VBX_REQUIRE(ctx, ctx && out_dptr, "null argument");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    VBX_HIP(ctx, hipMalloc(out_dptr, bytes ? bytes : 16));
    return VBX_SUCCESS;
}
int vbx_free(vbx_ctx *ctx, void *dptr) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (dptr) { VBX_HIP(ctx, hipStreamSynchronize(ctx->stream)); VBX_HIP(ctx, hipFree(dptr)); }
    return VBX_SUCCESS;
}
int vbx_memcpy_h2d(vbx_ctx *ctx, void *dst, const void *h_src, size_t bytes) {
    VBX_REQUIRE(ctx, ctx && (bytes == 0 || (dst && h_src)), "null argument");
    if (bytes == 0) return VBX_SUCCESS;
    VBX_HIP(ctx, hipMemcpyAsync(dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    VBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VBX_SUCCESS;
}
int vbx_memcpy_d2h(vbx_ctx *ctx, void *h_dst, const void *src, size_t bytes) {
    VBX_REQUIRE(ctx, ctx && (bytes == 0 || (h_dst && src)), "null argument");
    if (bytes == 0) return VBX_SUCCESS;
    VBX_HIP(ctx, hipMemcpyAsync(h_dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VBX_SUCCESS;
}
int vbx_memset(vbx_ctx *ctx, void *dst, int value, size_t bytes) {
    VBX_REQUIRE(ctx, ctx && (bytes == 0 || dst), "null argument");
    if (bytes == 0) return VBX_SUCCESS;
    VBX_HIP(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return VBX_SUCCESS;
}

int vbx_timer_begin(vbx_ctx *ctx) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    VBX_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
    return VBX_SUCCESS;
}
int vbx_timer_end(vbx_ctx *ctx, float *h_ms) {
    VBX_REQUIRE(ctx, ctx && h_ms, "null argument");
    VBX_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
    VBX_HIP(ctx, hipEventSynchronize(ctx->t1));
    VBX_HIP(ctx, hipEventElapsedTime(h_ms, ctx->t0, ctx->t1));
    return VBX_SUCCESS;
}

static int prof_flush(vbx_ctx *ctx) {
    if (ctx->recs.empty()) return VBX_SUCCESS;
    VBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->side) VBX_HIP(ctx, hipStreamSynchronize(ctx->side));
    if (ctx->trk) VBX_HIP(ctx, hipStreamSynchronize(ctx->trk));
    for (auto &r : ctx->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto &acc = ctx->prof_acc[r.name];
            acc.first += ms; acc.second += 1;
            ctx->prof_stream[r.name] = (r.stream == ctx->stream) ? 0 : (r.stream == ctx->side) ? 1 : 2;
        }
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    ctx->recs.clear();
    return VBX_SUCCESS;
}
int vbx_profile_enable(vbx_ctx *ctx, int on) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    int rc = prof_flush(ctx);
    ctx->prof = on != 0;
    return rc;
}
int vbx_profile_reset(vbx_ctx *ctx) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    int rc = prof_flush(ctx);
    ctx->prof_acc.clear();
    ctx->prof_stream.clear();
    if (ctx->pitch_work)
        VBX_HIP(ctx, hipMemsetAsync(ctx->pitch_work, 0, PITCH_WORK_WORDS * sizeof(unsigned long long), ctx->stream));
    return rc;
}
int vbx_profile_pitch_work(vbx_ctx *ctx, uint64_t *h_out4) {
    VBX_REQUIRE(ctx, ctx && h_out4, "null argument");
    for (int i = 0; i < 4; i++) h_out4[i] = 0;
    if (!ctx->pitch_work) return VBX_SUCCESS;
    unsigned long long h[PITCH_WORK_SLOTS * 4];
    VBX_HIP(ctx, hipMemcpyAsync(h, ctx->pitch_work, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    VBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < PITCH_WORK_SLOTS; s++) for (int i = 0; i < 4; i++) h_out4[i] += h[4 * s + i];
#ifdef VBX_EXP_PHASES
    {   // experiment build: the phase clock sums, to stderr
        unsigned long long ph[PITCH_WORK_SLOTS * PHASE_SLOTS], tot[PHASE_SLOTS] = {0};
        VBX_HIP(ctx, hipMemcpy(ph, ctx->pitch_work + PITCH_WORK_SLOTS * 4, sizeof ph, hipMemcpyDeviceToHost));
        for (int s = 0; s < PITCH_WORK_SLOTS; s++) for (int k = 0; k < PHASE_SLOTS; k++) tot[k] += ph[s * PHASE_SLOTS + k];
        fprintf(stderr, "VBX_PHASES frames %llu cycles", (unsigned long long)h_out4[0]);
        for (int k = 0; k < PHASE_SLOTS; k++) fprintf(stderr, " %llu", tot[k]);
        fprintf(stderr, "\n");
    }
#endif
    return VBX_SUCCESS;
}
int vbx_profile_get(vbx_ctx *ctx, const char *kernel_name, double *h_total_ms, long *h_launches) {
    VBX_REQUIRE(ctx, ctx && kernel_name, "null argument");
    int rc = prof_flush(ctx);
    if (rc != VBX_SUCCESS) return rc;
    auto it = ctx->prof_acc.find(kernel_name);
    if (h_total_ms) *h_total_ms = (it == ctx->prof_acc.end()) ? 0.0 : it->second.first;
    if (h_launches) *h_launches = (it == ctx->prof_acc.end()) ? 0 : it->second.second;
    return VBX_SUCCESS;
}
int vbx_profile_stream(vbx_ctx *ctx, const char *kernel_name, int *h_stream) {
    VBX_REQUIRE(ctx, ctx && kernel_name && h_stream, "null argument");
    int rc = prof_flush(ctx);
    if (rc != VBX_SUCCESS) return rc;
    auto it = ctx->prof_stream.find(kernel_name);
    *h_stream = (it == ctx->prof_stream.end()) ? -1 : it->second;
    return VBX_SUCCESS;
}
int vbx_profile_names(vbx_ctx *ctx, char *h_buf, size_t cap) {
    VBX_REQUIRE(ctx, ctx && h_buf && cap, "null argument");
    int rc = prof_flush(ctx);
    if (rc != VBX_SUCCESS) return rc;
    std::string s;
    for (auto &kv : ctx->prof_acc) { s += kv.first; s += '\n'; }
    std::strncpy(h_buf, s.c_str(), cap - 1); h_buf[cap - 1] = 0;
    return VBX_SUCCESS;
}

// ---- tables -------------------------------------------------------------------------------

// ---- periodic.rs --------------------------------------------------------------------------

// Autocorrelate::autocorrelate(n_lags) of every frame on stream st: the few-lag register kernel, one FFT of the zero-padded
// frame (many lags of a 512..4096-sample frame), or the matrix-core tiles.
static int run_autocorrelate(vbx_ctx *ctx, hipStream_t st, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                             const double *window, size_t n_lags, double *out) {
    if (frame_len > VBX_MAX_FRAME_LEN) {
        // a frame that no wavefront's LDS image holds: the matrix-core lag tiles over chunked images (k_long.hip)
        void *w = nullptr;
        int rc = ws_get(ctx, vbx_ctx::WS_LONG, autocorr_long_scratch_bytes((long)n_frames, (long)frame_len, (long)n_lags), &w);
        if (rc != VBX_SUCCESS) return rc;
        Prof p(ctx, "autocorr_long", st);
        launch_autocorr_long(st, x, (long)n_frames, (long)frame_len, (long)stride, window, (long)n_lags, out, (double *)w);
        return VBX_SUCCESS;
    }
    if (fewlags_supported((int)frame_len, (int)n_lags, false)) {
        Prof p(ctx, "autocorr_fewlags", st);
        launch_autocorr_fewlags(st, x, (long)n_frames, (int)frame_len, (long)stride, window, (int)n_lags, 0, out, nullptr);
    } else if (!ctx->pitch_force_mfma && spectral_plan((int)frame_len) != SPECTRAL_PLAN_NONE &&
               (n_lags >= SPECTRAL_AC_MIN_LAGS || frame_len >= 1024)) {
        // every lag sum from one real FFT of the zero-padded frame (k_spectral*.hip) instead of lags x frame_len products;
        // the rounding error, ~1e-16 r[0] per lag, is a thousandth of the tolerance's floor
        const double *tab = nullptr;
        spectral_launch_t L{};
        L.plan = spectral_plan((int)frame_len); L.n = (int)frame_len;
        int rc = get_spectral_tab(ctx, L.plan, &tab);
        if (rc != VBX_SUCCESS) return rc;
        L.x = x; L.F = (long)n_frames; L.stride = (long)stride; L.window = window; L.tab = tab;
        L.out_r = out; L.n_lags = (int)n_lags;
        Prof p(ctx, "autocorr_fft", st);
        launch_analyze(st, L);
    } else {
        Prof p(ctx, "autocorr_tiles", st);
        launch_autocorr_tiles(st, x, (long)n_frames, (int)frame_len, (long)stride, window, (int)n_lags, out);
    }
    return VBX_SUCCESS;
}

int vbx_autocorrelate_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len,
                          size_t stride, const double *window, size_t n_lags, double *out) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride, VBX_MAX_LONG_FRAME_LEN);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr, "null output");
    VBX_REQUIRE(ctx, n_lags >= 1 && n_lags <= frame_len, "n_lags must be in [1, frame_len] (the reference panics beyond)");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    rc = run_autocorrelate(ctx, ctx->stream, x, n_frames, frame_len, stride, window, n_lags, out);
    if (rc != VBX_SUCCESS) return rc;
    return check_launch(ctx, __func__);
}

int vbx_normalize_f64(vbx_ctx *ctx, double *data, size_t n_rows, size_t n) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_rows == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, data && n >= 1 && n <= 0x7fffffff && n_rows <= 0x7fffffff, "bad argument");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "normalize_rows"); launch_normalize_rows(ctx->stream, data, (long)n_rows, (int)n); }
    return check_launch(ctx, __func__);
}

int vbx_interpolate_sinc_f64(vbx_ctx *ctx, const double *y, size_t ylen, long offset, size_t nx,
                             const double *xs, size_t m, size_t max_depth, double *out, int32_t *status) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (m == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, y && xs && out && ylen >= 1 && ylen <= 0x7fffffff && m <= 0x7fffffff, "bad argument");
    VBX_REQUIRE(ctx, max_depth <= 0x3fffffff && nx <= 0x3fffffff && ylen <= 0x3fffffff &&
                offset > -0x3fffffffL && offset < 0x3fffffffL, "depth / nx / offset / ylen too large");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "sinc_points"); launch_sinc_points(ctx->stream, y, (int)ylen, offset, (long)nx, xs, (long)m, (long)max_depth, out, status); }
    return check_launch(ctx, __func__);
}

int vbx_improve_extremum_ex_f64(vbx_ctx *ctx, const double *y, size_t ylen, long offset, size_t nx,
                                const double *ixmid, size_t m, int interpolation, size_t depth, int is_max,
                                double *out_xy, int32_t *status) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (m == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, y && ixmid && out_xy && ylen >= 1 && ylen <= 0x7fffffff && m <= 0x7fffffff, "bad argument");
    VBX_REQUIRE(ctx, interpolation >= VBX_INTERP_NONE && interpolation <= VBX_INTERP_SINC, "interpolation must be NONE, PARABOLIC or SINC");
    VBX_REQUIRE(ctx, depth <= 0x3fffffff && nx <= 0x3fffffff && ylen <= 0x3fffffff &&
                offset > -0x3fffffffL && offset < 0x3fffffffL, "depth / nx / offset / ylen too large");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "extremum_points"); launch_extremum_points(ctx->stream, y, (int)ylen, offset, (long)nx, ixmid, (long)m, (long)depth, out_xy, status, interpolation, is_max ? 1 : 0); }
    return check_launch(ctx, __func__);
}

int vbx_improve_extremum_f64(vbx_ctx *ctx, const double *y, size_t ylen, long offset, size_t nx,
                             const double *ixmid, size_t m, size_t depth, double *out_xy, int32_t *status) {
    return vbx_improve_extremum_ex_f64(ctx, y, ylen, offset, nx, ixmid, m, VBX_INTERP_SINC, depth, 1, out_xy, status);
}

// The FFT-based kernel (k_spectral.hip) followed by the direct-sum kernel on the frames whose peak decisions lie
// inside the FFT's rounding error (normally none; curves that are exactly zero over a stretch, e.g. a few impulses).
static int launch_spectral(vbx_ctx *ctx, hipStream_t st, spectral_launch_t &L, const char *prof_name) {
    void *w = nullptr;
    int rc = ws_get(ctx, vbx_ctx::WS_UNSURE, ((size_t)L.F + 4) * sizeof(int32_t), &w);
    if (rc != VBX_SUCCESS) return rc;
    L.unsure_count = (int32_t *)w;                       // [0]: count, [4..]: frame indices
    L.unsure_list = (int32_t *)w + 4;
    VBX_HIP(ctx, hipMemsetAsync(L.unsure_count, 0, sizeof(int32_t), st));
    // LPC rows: levinson_rows_kernel_t (k_lpc.hip) lists the frames whose Levinson row a few eps of lag-sum rounding can move by more
    // than 1e-6; lpc_exact_list_kernel redoes those in double-double.  VBX_LPC_EXACT=0: no probe, the rows of rounds 1-5 (A/B, tests).
    L.lpc_list = nullptr; L.lpc_count = nullptr;
    if (L.out_lpc != nullptr && ctx->lpc_exact) {
        void *lw = nullptr;
        rc = ws_get(ctx, vbx_ctx::WS_LPC_LIST, ((size_t)L.F + 4) * sizeof(int32_t), &lw);
        if (rc != VBX_SUCCESS) return rc;
        L.lpc_count = (int32_t *)lw; L.lpc_list = (int32_t *)lw + 4;
        VBX_HIP(ctx, hipMemsetAsync(L.lpc_count, 0, sizeof(int32_t), st));
    }
    // the 4096-point plan runs as two kernels with the lag curves in a scratch buffer between them (vbx_spectral.hpp, SP_ANALYZE_SPLIT):
    // batches of up to 131,072 frames (~10 KB each)
    L.curve_ws = nullptr; L.curve_ws_bytes = 0;
    const size_t rowb = (ctx->pow2_split != 0 && !L.whole_curve && !L.mfcc_only && L.out_r == nullptr && pitch_full_list_bytes(L.n, L.kmax) == 0)
                            ? spectral_split_row_bytes(L.n, L.sample_rate, L.fmin) : 0;      // (a list region in LDS, kmax > 64: the fused kernel)
    if (rowb) {
        const size_t frames = (size_t)L.F < 131072 ? (size_t)L.F : 131072;
        const size_t need = (frames < 1024 ? 1024 : frames) * rowb + 64;
        void *cw = nullptr;
        // (no memory for it: the fused form.  A size that failed once is not asked for again on every call -- each attempt drains the
        // streams and fails a multi-GB hipMalloc --, and the fallback is not an error of this call: the error string is put back)
        if (ctx->curve_failed_bytes == 0 || need < ctx->curve_failed_bytes) {
            const std::string before = ctx->last_error;
            if (ws_get(ctx, vbx_ctx::WS_CURVE, need, &cw) == VBX_SUCCESS) { L.curve_ws = (double *)cw; L.curve_ws_bytes = need; }
            else { (void)hipGetLastError(); ctx->curve_failed_bytes = need; ctx->last_error = before; g_last_error = before; }
        }
    }
    // MFCC::mfcc's log10 + DCT out of the frame's wavefront (mfcc_tail_q's `defer`): the kernel leaves the filter sums in the row
    // (the 1200-point plan only: in the power-of-two kernels the deferred form changes the register allocation -- thirteen index registers
    // spilled across the second transform, 3 KB of scratch traffic per frame -- for the same ~1 %; they keep the tail)
    L.mfcc_defer = L.out_mfcc != nullptr && !L.mfcc_only && L.num_coeffs >= 1 && L.num_coeffs <= 16 && ctx->mfcc_defer && L.plan == SPECTRAL_PLAN_1200;
    { Prof p(ctx, prof_name, st); ctx->last_spectral_split = launch_analyze(st, L); }
    if (L.mfcc_defer && L.out_lpc == nullptr) { Prof p(ctx, "mfcc_rows", st); launch_mfcc_rows(st, L.out_mfcc, L.F, L.mfcc_ld, L.num_coeffs, L.dct); }
    {
        Prof p(ctx, "pitch_direct_fallback", st);
        // a fixed grid over a count only the device knows (almost always zero).  Every workgroup of this kernel allocates the
        // frame's LDS image + refinement state before it can look at the count: ~70 KB at 3000 samples -- 1024 of them took
        // 4 ms to come and go with nothing to do, beside 13.5 ms of the analysis itself.  One per CU where the state is large.
        const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
        const int grid = pitch_lds_bytes(L.n) > 24 * 1024 ? cus : cus * 4;
        launch_pitch_list(st, L.unsure_list, L.unsure_count, grid, L.x, L.n, L.stride, L.window, L.lag_window,
                          L.sample_rate, L.threshold, L.fmin, L.fmax, L.kmax, L.out_cand, L.cand_ld, L.out_count,
                          L.pitch_status, L.work, L.pcm);
    }
    if (L.out_lpc != nullptr) {
        // the fused kernel left r[0..12] in every frame's LPC row: LPC::lpc(12) in place, one row per lane (+ the conditioning probe), then
        // the listed rows again from their frames in double-double
        { Prof p(ctx, "lpc_rows", st);
          launch_levinson_rows_probe(st, L.out_lpc, L.F, L.lpc_ld, SPECTRAL_LPC_ORDER, L.out_lpc, L.lpc_ld, L.lpc_list, L.lpc_count,
                                     L.mfcc_defer ? L.out_mfcc : nullptr, L.mfcc_ld, L.num_coeffs, L.dct); }        // (+ the deferred MFCC tail of the same record)
        if (L.lpc_list != nullptr) {
            Prof p(ctx, "lpc_exact_list", st);
            const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
            launch_lpc_exact_list(st, L.lpc_list, L.lpc_count, cus, L.x, L.n, L.stride, L.window, L.pcm, SPECTRAL_LPC_ORDER, L.out_lpc, L.lpc_ld);
        }
    }
    return check_launch(ctx, "launch_spectral");
}

static int run_pitch(vbx_ctx *ctx, hipStream_t st, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                     const double *window, double sample_rate, double threshold, double fmin, double fmax,
                     size_t kmax, vbx_pitch *out_cand, size_t cand_ld, int32_t *out_count, int32_t *status) {
    VBX_REQUIRE(ctx, out_cand != nullptr, "null output");
    const size_t kcap = frame_len > VBX_MAX_FRAME_LEN ? VBX_PITCH_MAX_CANDIDATES(frame_len) : (size_t)VBX_MAX_PITCH_CANDIDATES;
    VBX_REQUIRE(ctx, kmax >= 1 && kmax <= kcap, "kmax must be in [1, VBX_MAX_PITCH_CANDIDATES] (long frames: [1, frame_len / 4 + 2])");
    VBX_REQUIRE(ctx, cand_ld >= 2 * kmax && cand_ld % 2 == 0, "candidate rows must be 16-byte aligned and hold kmax entries");
    VBX_REQUIRE(ctx, frame_len >= 4, "frame_len must be >= 4");
    const double *lagw = nullptr;
    int rc = get_window_dev(ctx, VBX_WINDOW_HANNING_LAG, frame_len, &lagw);
    if (rc != VBX_SUCCESS) return rc;
    if (frame_len > VBX_MAX_FRAME_LEN) {
        // a frame whose lag curve no LDS holds (k_long.hip): every lag by the chunked matrix-core tiles, the curve as an array in
        // HBM, peak scan -> improve_extremum per candidate -> rank sort; batches of frames so that the scratch stays <= 2 GiB
        VBX_REQUIRE(ctx, frame_len <= 0x3fffffff, "frame_len too large for the lag curve's 32-bit indices");
        size_t per = (size_t(1) << 31) / pitch_long_scratch_bytes(1, (long)frame_len);
        if (per < 1) per = 1;
        if (per > n_frames) per = n_frames;
        void *w2 = nullptr;
        rc = ws_get(ctx, vbx_ctx::WS_LONG2, pitch_long_scratch_bytes((long)per, (long)frame_len), &w2);
        if (rc != VBX_SUCCESS) return rc;
        for (size_t f0 = 0; f0 < n_frames; f0 += per) {
            const size_t m = (n_frames - f0 < per) ? n_frames - f0 : per;
            double *r = pitch_long_r(w2);
            rc = run_autocorrelate(ctx, st, x + f0 * stride, m, frame_len, stride, window, frame_len, r);      // :402
            if (rc != VBX_SUCCESS) return rc;
            { Prof p(ctx, "normalize_rows", st); launch_normalize_rows(st, r, (long)m, (int)frame_len); }       // :404
            { Prof p(ctx, "pitch_long", st);
              launch_pitch_long(st, (long)m, (long)frame_len, lagw, sample_rate, threshold, fmin, fmax, (int)kmax,
                                (double *)out_cand + f0 * cand_ld, (long)cand_ld, out_count ? out_count + f0 : nullptr,
                                status ? status + f0 : nullptr, w2); }
        }
        return check_launch(ctx, "vbx_pitch_f64");
    }
    VBX_REQUIRE(ctx, pitch_lds_bytes((int)frame_len) + pitch_full_list_bytes((int)frame_len, (int)kmax) + 16 <= 160 * 1024,
                "frame does not fit the LDS");
    if (ctx->prof && !ctx->pitch_work) {
        const size_t wb = PITCH_WORK_WORDS * sizeof(unsigned long long);
        VBX_HIP(ctx, hipMalloc((void **)&ctx->pitch_work, wb));
        VBX_HIP(ctx, hipMemsetAsync(ctx->pitch_work, 0, wb, st));
    }
    if (!ctx->pitch_force_mfma && spectral_supported((int)frame_len, 0, 0, 0, 0)) {
        // autocorrelation by one real FFT of the frame (k_spectral.hip) instead of the O(N^2) lag sums
        const double *tab = nullptr;
        spectral_launch_t L{};
        L.plan = spectral_plan((int)frame_len); L.n = (int)frame_len;
        rc = get_spectral_tab(ctx, L.plan, &tab);
        if (rc != VBX_SUCCESS) return rc;
        L.x = x; L.F = (long)n_frames; L.stride = (long)stride; L.window = window; L.lag_window = lagw; L.tab = tab;
        L.lag_rcp = ctx->lag_rcp_ok.count(frame_len) && ctx->lag_rcp_ok[frame_len];
        L.sample_rate = sample_rate; L.threshold = threshold; L.fmin = fmin; L.fmax = fmax; L.kmax = (int)kmax;
        L.whole_curve = ctx->pitch_whole_curve;
        L.out_cand = (pitch_t *)out_cand; L.cand_ld = (long)cand_ld; L.out_count = out_count; L.pitch_status = status;
        L.work = ctx->prof ? ctx->pitch_work : nullptr;
        return launch_spectral(ctx, st, L, "pitch");
    }
    {
        Prof p(ctx, "pitch", st);
        launch_pitch(st, x, (long)n_frames, (int)frame_len, (long)stride, window, lagw, sample_rate, threshold,
                     fmin, fmax, (int)kmax, (pitch_t *)out_cand, (long)cand_ld, out_count, status,
                     ctx->prof ? ctx->pitch_work : nullptr);
    }
    return check_launch(ctx, "vbx_pitch_f64");
}

int vbx_pitch_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                  const double *window, double sample_rate, double threshold, double fmin, double fmax,
                  size_t kmax, vbx_pitch *out_cand, int32_t *out_count, int32_t *status) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride, VBX_MAX_LONG_FRAME_LEN);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    return run_pitch(ctx, ctx->stream, x, n_frames, frame_len, stride, window, sample_rate, threshold, fmin, fmax,
                     kmax, out_cand, 2 * kmax, out_count, status);
}

// ---- spectrum.rs: LPC ---------------------------------------------------------------------

int vbx_lpc_mut_f64(vbx_ctx *ctx, const double *r, size_t n_frames, size_t r_stride, size_t n_coeffs, double *out_ac,
                    double *out_kc) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_frames == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, r && out_ac, "null argument");
    VBX_REQUIRE(ctx, n_coeffs >= 1 && n_coeffs <= VBX_MAX_LPC_ORDER && r_stride >= n_coeffs + 1, "bad order / stride");
    VBX_REQUIRE(ctx, n_frames <= 0x7fffffffull, "too many rows");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "levinson_rows"); launch_levinson_rows(ctx->stream, r, (long)n_frames, (long)r_stride, (int)n_coeffs, out_ac, (long)n_coeffs + 1, out_kc); }
    return check_launch(ctx, __func__);
}

int vbx_lpc_f64(vbx_ctx *ctx, const double *r, size_t n_frames, size_t r_stride, size_t n_coeffs, double *out) {
    return vbx_lpc_mut_f64(ctx, r, n_frames, r_stride, n_coeffs, out, nullptr);
}

static int run_autocorr_lpc(vbx_ctx *ctx, hipStream_t st, const double *x, size_t n_frames, size_t frame_len,
                            size_t stride, const double *window, size_t n_coeffs, int normalize,
                            double *out_r, double *out_lpc, size_t lpc_ld) {
    VBX_REQUIRE(ctx, out_r || out_lpc, "both outputs null");
    VBX_REQUIRE(ctx, n_coeffs >= 1 && n_coeffs <= VBX_MAX_LPC_ORDER && n_coeffs + 1 <= frame_len, "bad order");
    VBX_REQUIRE(ctx, lpc_ld >= n_coeffs + 1, "LPC rows must hold n_coeffs + 1 entries");
    const int n_lags = (int)n_coeffs + 1;
    int rc;
    // LPC rows: a conditioning probe lists the frames whose row a few eps of lag-sum rounding can move by more than 1e-6; those are
    // redone from the frame in double-double (k_lpc_exact.hip).  VBX_LPC_EXACT=0: the rows of rounds 1-5.
    int32_t *lpc_list = nullptr, *lpc_count = nullptr;
    if (out_lpc != nullptr && ctx->lpc_exact && lpc_exact_supported((int)frame_len, (int)n_coeffs)) {
        void *lw = nullptr;
        rc = ws_get(ctx, vbx_ctx::WS_LPC_LIST, (n_frames + 4) * sizeof(int32_t), &lw);
        if (rc != VBX_SUCCESS) return rc;
        lpc_count = (int32_t *)lw; lpc_list = (int32_t *)lw + 4;
        VBX_HIP(ctx, hipMemsetAsync(lpc_count, 0, sizeof(int32_t), st));
    }
    auto redo = [&]() {
        if (lpc_list == nullptr) return;
        Prof p(ctx, "lpc_exact_list", st);
        const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
        launch_lpc_exact_list(st, lpc_list, lpc_count, cus, x, (int)frame_len, (long)stride, window, false, (int)n_coeffs, out_lpc, (long)lpc_ld);
    };
    if (fewlags_supported((int)frame_len, n_lags, out_lpc != nullptr)) {
        { Prof p(ctx, "autocorr_lpc", st);
          launch_autocorr_fewlags(st, x, (long)n_frames, (int)frame_len, (long)stride, window, n_lags, normalize, out_r, out_lpc, (long)lpc_ld,
                                  lpc_list, lpc_count); }
        redo();
        return check_launch(ctx, "vbx_autocorr_lpc_f64");
    }
    // general shapes: autocorrelate -> [normalize] -> Levinson as three launches
    double *r = out_r;
    if (!r) {
        void *w = nullptr;
        rc = ws_get(ctx, vbx_ctx::WS_MISC, n_frames * (size_t)n_lags * sizeof(double), &w);
        if (rc != VBX_SUCCESS) return rc;
        r = (double *)w;
    }
    rc = run_autocorrelate(ctx, st, x, n_frames, frame_len, stride, window, (size_t)n_lags, r);
    if (rc != VBX_SUCCESS) return rc;
    if (normalize) { Prof p(ctx, "normalize_rows", st); launch_normalize_rows(st, r, (long)n_frames, n_lags); }
    if (out_lpc) {
        Prof p(ctx, "levinson_rows", st);
        if (lpc_list != nullptr) launch_levinson_rows_probe(st, r, (long)n_frames, n_lags, (int)n_coeffs, out_lpc, (long)lpc_ld, lpc_list, lpc_count);
        else launch_levinson_rows(st, r, (long)n_frames, n_lags, (int)n_coeffs, out_lpc, (long)lpc_ld);
    }
    redo();
    return check_launch(ctx, "vbx_autocorr_lpc_f64");
}

int vbx_autocorr_lpc_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len,
                         size_t stride, const double *window, size_t n_coeffs, int normalize,
                         double *out_r, double *out_lpc) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride, VBX_MAX_LONG_FRAME_LEN);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    return run_autocorr_lpc(ctx, ctx->stream, x, n_frames, frame_len, stride, window, n_coeffs, normalize, out_r, out_lpc, n_coeffs + 1);
}

static bool burg_order_ok(size_t frame_len, size_t n_coeffs) {
    return frame_len >= 2 && n_coeffs >= 1 && n_coeffs <= VBX_MAX_LPC_ORDER;
}

// Burg on a batch: the one-pass form (k_burg_fast.hip) where it exists, the frames its guard turns away and every other
// shape through the direct recursion (k_burg.hip)
static int run_burg(vbx_ctx *ctx, hipStream_t stm, const double *x, const int16_t *pcm, long F, int n, long stride,
                    const double *window, int p, double *coeffs, int32_t *st, frame_map_t map = frame_map_t{0, 0, 0}) {
    if (n > VBX_MAX_FRAME_LEN) {
        // a frame longer than a wavefront's registers hold (tests/lib.rs:27-41 passes a whole file as one): one workgroup per
        // frame, the error arrays in an L2-resident scratch (k_long.hip); batches of frames so that the scratch stays <= 1 GiB
        ctx->burg_list_count = nullptr;
        long per = (long)((size_t(1) << 30) / burg_long_scratch_bytes(1, n));
        if (per < 1) per = 1;
        if (per > F) per = F;
        void *w = nullptr;
        int rc = ws_get(ctx, vbx_ctx::WS_LONG, burg_long_scratch_bytes(per, n), &w);
        if (rc != VBX_SUCCESS) return rc;
        Prof pr(ctx, "burg_long", stm);
        for (long f0 = 0; f0 < F; f0 += per)
            launch_burg_long(stm, x, f0, (f0 + per < F) ? f0 + per : F, F, n, stride, window, p, coeffs, st, (double *)w);
        return VBX_SUCCESS;
    }
    if (burg_fast_supported(n, p)) {
        void *w = nullptr;
        int rc = ws_get(ctx, vbx_ctx::WS_BURG_LIST, burg_fast_scratch_bytes(F, p), &w);
        if (rc != VBX_SUCCESS) return rc;
        int32_t *list = burg_fast_list(w, F, p);
        ctx->burg_list_count = list;
        VBX_HIP(ctx, hipMemsetAsync(list, 0, sizeof(int32_t), stm));
        const long items = frame_map_items(map, F), chunk = burg_fast_chunk(F);
        for (long i0 = 0; i0 < items; i0 += chunk) {
            const long m = (items - i0 < chunk) ? items - i0 : chunk;
            { Prof pr(ctx, "burg_lags", stm);
              if (pcm) launch_burg_lags_pcm16(stm, pcm, F, n, stride, window, p, map, i0, m, w);
              else launch_burg_lags(stm, x, F, n, stride, window, p, map, i0, m, w); }
            { Prof pr(ctx, "burg_recursion", stm); launch_burg_recursion(stm, F, p, map, i0, m, coeffs, st, w); }
        }
        { Prof pr(ctx, "burg_direct_list", stm);
          if (pcm) launch_burg_pcm16_list(stm, pcm, F, n, stride, window, p, coeffs, st, list + 2, list);
          else launch_burg_list(stm, x, F, n, stride, window, p, coeffs, st, list + 2, list); }
        // the probe (vbx_internal_last_burg_direct_count) reports the whole CALL: list[0] restarts with every time slice of
        // find_formants, list[1] adds the slices up (reset with the call's first slice)
        launch_count_accumulate(stm, list, list + 1, map.seg_len == 0 || map.t0 == 0);
        return VBX_SUCCESS;
    }
    Prof pr(ctx, "burg", stm);
    ctx->burg_list_count = nullptr;
    if (pcm) launch_burg_pcm16(stm, pcm, F, n, stride, window, p, coeffs, st, map);
    else launch_burg(stm, x, F, n, stride, window, p, coeffs, st, map);
    return VBX_SUCCESS;
}

int vbx_lpc_burg_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len,
                     size_t stride, const double *window, size_t n_coeffs, double *out, int32_t *status) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride, VBX_MAX_LONG_FRAME_LEN);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr, "null output");
    VBX_REQUIRE(ctx, burg_order_ok(frame_len, n_coeffs), "frame_len must be >= 2, order in [1, 62]");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    rc = run_burg(ctx, ctx->stream, x, nullptr, (long)n_frames, (int)frame_len, (long)stride, window, (int)n_coeffs, out, status);
    if (rc != VBX_SUCCESS) return rc;
    return check_launch(ctx, __func__);
}

// ---- polynomial.rs ------------------------------------------------------------------------

int vbx_find_roots_c64(vbx_ctx *ctx, vbx_complex *polys, size_t n_polys, size_t len, int32_t *status) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_polys == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, polys != nullptr, "null polynomials");
    VBX_REQUIRE(ctx, len >= 1 && len <= VBX_MAX_POLY_LEN, "len must be in [1, 64]");
    VBX_REQUIRE(ctx, n_polys <= 0x7fffffffull, "too many polynomials");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "find_roots"); launch_find_roots(ctx->stream, (cplx_t *)polys, (long)n_polys, (int)len, status); }
    return check_launch(ctx, __func__);
}

int vbx_laguerre_c64(vbx_ctx *ctx, const vbx_complex *polys, size_t n_polys, size_t len,
                     vbx_complex start, vbx_complex *out) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_polys == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, polys && out, "null argument");
    VBX_REQUIRE(ctx, len >= 2 && len <= VBX_MAX_POLY_LEN, "len must be in [2, 64]");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    cplx_t s; s.re = start.re; s.im = start.im;
    { Prof p(ctx, "laguerre"); launch_laguerre(ctx->stream, (const cplx_t *)polys, (long)n_polys, (int)len, s, (cplx_t *)out); }
    return check_launch(ctx, __func__);
}

int vbx_div_polynomial_c64(vbx_ctx *ctx, vbx_complex *polys, const vbx_complex *others, size_t n_polys, size_t len,
                           vbx_complex *rem, int32_t *status) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_polys == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, polys && others && rem, "null argument");
    VBX_REQUIRE(ctx, len >= 1 && len <= VBX_MAX_POLY_LEN, "len must be in [1, 64]");
    VBX_REQUIRE(ctx, n_polys <= 0x7fffffffull, "too many polynomials");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "div_polynomial"); launch_div_polynomial(ctx->stream, (cplx_t *)polys, (const cplx_t *)others, (long)n_polys, (int)len, (cplx_t *)rem, status); }
    return check_launch(ctx, __func__);
}

int vbx_find_roots_c32(vbx_ctx *ctx, vbx_complex32 *polys, size_t n_polys, size_t len, int32_t *status) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_polys == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, polys != nullptr, "null polynomials");
    VBX_REQUIRE(ctx, len >= 1 && len <= VBX_MAX_POLY_LEN, "len must be in [1, 64]");
    VBX_REQUIRE(ctx, n_polys <= 0x7fffffffull, "too many polynomials");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "find_roots_f32"); launch_find_roots_f32(ctx->stream, (cplx32_t *)polys, (long)n_polys, (int)len, status); }
    return check_launch(ctx, __func__);
}

int vbx_laguerre_c32(vbx_ctx *ctx, const vbx_complex32 *polys, size_t n_polys, size_t len,
                     vbx_complex32 start, vbx_complex32 *out) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_polys == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, polys && out, "null argument");
    VBX_REQUIRE(ctx, len >= 2 && len <= VBX_MAX_POLY_LEN, "len must be in [2, 64]");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    cplx32_t s; s.re = start.re; s.im = start.im;
    { Prof p(ctx, "laguerre_f32"); launch_laguerre_f32(ctx->stream, (const cplx32_t *)polys, (long)n_polys, (int)len, s, (cplx32_t *)out); }
    return check_launch(ctx, __func__);
}

// ---- spectrum.rs: resonances, tracker -----------------------------------------------------

int vbx_to_resonance_c64(vbx_ctx *ctx, const vbx_complex *roots, size_t n_rows, size_t n_roots,
                         double sample_rate, vbx_resonance *out_res, int32_t *out_count) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_rows == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, roots && out_res && n_roots >= 1 && n_roots <= 0x7fffffff, "bad argument");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    {
        Prof p(ctx, "to_resonance");
        launch_to_resonance(ctx->stream, (const cplx_t *)roots, (long)n_rows, (int)n_roots, sample_rate, 0,
                            (res_t *)out_res, (int)n_roots, out_count, nullptr);
    }
    return check_launch(ctx, __func__);
}

// Utterances of a few hundred frames or more take the chunked scan (k_tracker.hip: speculative chunks + exact repair, the
// same rows bit for bit, ~1 ms whatever the lengths): one lane per utterance costs ~5.4 us per frame of the LONGEST
// utterance.  VBX_TRACKER_CHUNKED=1 / 0 forces / forbids it (tests, A/B runs).
static bool tracker_wants_chunks(const int64_t *h_seg_start, size_t n_segments, size_t n_frames) {
    const char *env = getenv("VBX_TRACKER_CHUNKED");          // read per call: tests switch it
    const int forced = env ? atoi(env) : -1;
    if (forced >= 0) return forced != 0;
    long longest = 0;
    if (h_seg_start == nullptr || n_segments == 0) longest = (long)n_frames;
    else for (size_t i = 0; i < n_segments; i++) {
        const long end = (i + 1 < n_segments) ? (long)h_seg_start[i + 1] : (long)n_frames;
        if (end - (long)h_seg_start[i] > longest) longest = end - (long)h_seg_start[i];
    }
    // 2 ms of scan on one lane, where the chunked scan takes about 0.6 for any batch; and large batches of SHORT utterances
    // too since round 3: with Burg in one pass and the roots from conjugate pairs nothing is left for the time slices of
    // run_find_formants to hide the sequential scan behind (a million frames in utterances of 64 / 256 / 383 frames: sliced
    // 2.15 / 2.21 / 2.27 ms, chunked 1.84 / 2.15 / 2.13)
    return longest >= 384 || n_frames >= 65536;
}

static int run_tracker(vbx_ctx *ctx, hipStream_t st, bool chunked, const res_t *res, long F, int n_res, const int32_t *res_count,
                       const int64_t *d_seg, long nseg, const res_t *d_est, int n_est, const int32_t *frame_status,
                       res_t *out, long out_ld) {
    if (!chunked) {
        Prof p(ctx, "tracker", st);
        launch_tracker(st, res, F, n_res, res_count, d_seg, nseg, d_est, n_est, frame_status, out, out_ld);
        return VBX_SUCCESS;
    }
    void *w = nullptr;
    int rc = ws_get(ctx, vbx_ctx::WS_TRK, tracker_chunked_workspace_bytes(F), &w);
    if (rc != VBX_SUCCESS) return rc;
    Prof p(ctx, "tracker_chunked", st);
    launch_tracker_chunked(st, res, F, n_res, res_count, d_seg, nseg, d_est, n_est, frame_status, out, out_ld, w);
    return VBX_SUCCESS;
}

int vbx_estimate_formants_f64(vbx_ctx *ctx, const vbx_resonance *res, size_t n_frames, size_t n_res,
                              const int64_t *h_seg_start, size_t n_segments,
                              const vbx_resonance *h_est_init, size_t n_est,
                              const int32_t *frame_status, vbx_resonance *out) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_frames == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, res && h_est_init && out, "null argument");
    VBX_REQUIRE(ctx, n_res >= 1 && n_res <= 0x7fffffff, "n_res must be >= 1 (the reference indexes resonances[0])");
    VBX_REQUIRE(ctx, n_est >= 1 && n_est <= VBX_FORMANT_SLOTS, "n_est must be in [1, 6]");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t *d_seg = nullptr; size_t nseg = 1; const res_t *d_est = nullptr;
    int rc = upload_segments(ctx, ctx->stream, h_seg_start, n_segments, n_frames, &d_seg, &nseg);
    if (rc != VBX_SUCCESS) return rc;
    rc = upload_estimates(ctx, ctx->stream, h_est_init, n_est, &d_est);
    if (rc != VBX_SUCCESS) return rc;
    rc = run_tracker(ctx, ctx->stream, tracker_wants_chunks(h_seg_start, n_segments, n_frames), (const res_t *)res, (long)n_frames,
                     (int)n_res, nullptr, d_seg, (long)nseg, d_est, (int)n_est, frame_status, (res_t *)out, 2 * (long)n_est);
    if (rc != VBX_SUCCESS) return rc;
    return check_launch(ctx, __func__);
}

// internal (tests only; not part of the public header): the same scan with the per-row counts find_formants keeps beside
// its resonance rows (rows of `count` real entries with ascending positive frequencies, then zeros, may take the tracker's
// index form: k_tracker.hip)
int vbx_internal_estimate_formants_counted(vbx_ctx *ctx, const vbx_resonance *res, size_t n_frames, size_t n_res,
                                           const int32_t *res_count, const int64_t *h_seg_start, size_t n_segments,
                                           const vbx_resonance *h_est_init, size_t n_est,
                                           const int32_t *frame_status, vbx_resonance *out) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_frames == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, res && res_count && h_est_init && out, "null argument");
    VBX_REQUIRE(ctx, n_res >= 1 && n_res <= 0x7fffffff && n_est >= 1 && n_est <= VBX_FORMANT_SLOTS, "bad size");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t *d_seg = nullptr; size_t nseg = 1; const res_t *d_est = nullptr;
    int rc = upload_segments(ctx, ctx->stream, h_seg_start, n_segments, n_frames, &d_seg, &nseg);
    if (rc != VBX_SUCCESS) return rc;
    rc = upload_estimates(ctx, ctx->stream, h_est_init, n_est, &d_est);
    if (rc != VBX_SUCCESS) return rc;
    rc = run_tracker(ctx, ctx->stream, tracker_wants_chunks(h_seg_start, n_segments, n_frames), (const res_t *)res, (long)n_frames,
                     (int)n_res, res_count, d_seg, (long)nseg, d_est, (int)n_est, frame_status, (res_t *)out, 2 * (long)n_est);
    if (rc != VBX_SUCCESS) return rc;
    return check_launch(ctx, __func__);
}

// Burg coefficients -> resonance rows: the conjugate-pair form (k_roots_fast.hip) where it exists (it does the frames that
// fail its own check again by the reference's iteration), every other order through the reference's iteration (k_roots.hip)
static int run_formant_resonances(vbx_ctx *ctx, hipStream_t stm, const double *coeffs, long F, int p, double sample_rate,
                                  res_t *res, int32_t *cnt, int32_t *st, frame_map_t map = frame_map_t{0, 0, 0}) {
    if (formant_resonances_fast_supported(p)) {
        void *w = nullptr;
        int rc = ws_get(ctx, vbx_ctx::WS_ROOTS_LIST, 4 * sizeof(int32_t), &w);
        if (rc != VBX_SUCCESS) return rc;
        int32_t *redo = (int32_t *)w;
        ctx->roots_list_count = redo;
        if (map.seg_len == 0 || map.t0 == 0) VBX_HIP(ctx, hipMemsetAsync(redo, 0, sizeof(int32_t), stm));   // once per call
        Prof pr(ctx, "formant_resonances", stm);
        launch_formant_resonances_fast(stm, coeffs, F, p, sample_rate, res, cnt, st, map, redo);
        return VBX_SUCCESS;
    }
    ctx->roots_list_count = nullptr;
    Prof pr(ctx, "formant_resonances", stm);
    launch_formant_resonances(stm, coeffs, F, p, sample_rate, res, cnt, st, map);
    return VBX_SUCCESS;
}

static int run_find_formants(vbx_ctx *ctx, hipStream_t stm, const double *x, size_t n_frames, size_t frame_len,
                             size_t stride, double sample_rate, size_t n_coeffs,
                             const int64_t *h_seg_start, size_t n_segments,
                             const vbx_resonance *h_est_init, size_t n_est,
                             vbx_resonance *out_formants, size_t formants_ld, vbx_resonance *out_res, int32_t *out_res_count,
                             double *out_coeffs, int32_t *status, const int16_t *pcm = nullptr /* the frames as 16-bit PCM instead of x */) {
    VBX_REQUIRE(ctx, h_est_init && out_formants, "null argument");
    VBX_REQUIRE(ctx, burg_order_ok(frame_len, n_coeffs), "frame_len must be >= 2, order in [1, 62]");
    VBX_REQUIRE(ctx, n_est >= 1 && n_est <= VBX_FORMANT_SLOTS, "n_est must be in [1, 6]");
    VBX_REQUIRE(ctx, formants_ld >= 2 * n_est && formants_ld % 2 == 0, "formant rows must be 16-byte aligned and hold n_est entries");
    const long F = (long)n_frames; const int p = (int)n_coeffs;
    int rc;
    void *w = nullptr;
    double *coeffs = out_coeffs;
    if (!coeffs) { rc = ws_get(ctx, vbx_ctx::WS_COEFFS, n_frames * n_coeffs * sizeof(double), &w); if (rc) return rc; coeffs = (double *)w; }
    res_t *res = (res_t *)out_res;
    if (!res) { rc = ws_get(ctx, vbx_ctx::WS_RES, n_frames * VBX_MAX_RESONANCES * sizeof(vbx_resonance), &w); if (rc) return rc; res = (res_t *)w; }
    int32_t *cnt = out_res_count;
    if (!cnt) { rc = ws_get(ctx, vbx_ctx::WS_COUNT, n_frames * sizeof(int32_t), &w); if (rc) return rc; cnt = (int32_t *)w; }
    int32_t *st = status;
    if (!st) { rc = ws_get(ctx, vbx_ctx::WS_STATUS, n_frames * sizeof(int32_t), &w); if (rc) return rc; st = (int32_t *)w; }
    const double *hann = nullptr;
    rc = get_window_dev(ctx, VBX_WINDOW_HANNING_PERIODIC, frame_len, &hann);   // src/lib.rs:65-70
    if (rc != VBX_SUCCESS) return rc;
    const int64_t *d_seg = nullptr; size_t nseg = 1; const res_t *d_est = nullptr;
    rc = upload_segments(ctx, stm, h_seg_start, n_segments, n_frames, &d_seg, &nseg);
    if (rc != VBX_SUCCESS) return rc;
    rc = upload_estimates(ctx, stm, h_est_init, n_est, &d_est);
    if (rc != VBX_SUCCESS) return rc;
    // The tracker is a chain of dependent steps per utterance (~5 us per frame, whatever the batch size).  Utterances long
    // enough for that to matter take the chunked scan after Burg and the root finder (run_tracker).  The alternative kept
    // behind VBX_TRACKER_CHUNKED=0 -- round 2's first answer, for batches of equal-length utterances: the work is cut into
    // time slices, and while the tracker walks frames [t0, t0 + tc) of every utterance on its own stream, Burg and the root
    // finder already produce the next slice (config 4 in round 2: 152 M frames/s against the chunked scan's 156 M).  Since
    // round 3 only VBX_TRACKER_CHUNKED=0 reaches it (tracker_wants_chunks).
    long seg_len = 0;
    const bool chunked = tracker_wants_chunks(h_seg_start, n_segments, n_frames);   // long utterances: the chunked scan instead
    if (!chunked && h_seg_start != nullptr && n_segments >= 64 && F >= 65536) {
        seg_len = (n_segments > 1) ? (long)h_seg_start[1] : 0;
        for (size_t i = 0; i < n_segments && seg_len > 0; i++) if (h_seg_start[i] != (int64_t)i * seg_len) seg_len = 0;
        // the slices cover t in [0, seg_len) of every utterance: a LAST utterance longer than the others (its end is
        // n_frames, not a seg_start entry) would keep rows past seg_len that no slice tracks -> the unsliced path
        if (seg_len > 0 && ((long)(n_segments - 1) * seg_len >= F || (long)n_segments * seg_len < F || seg_len < 64)) seg_len = 0;
    }
    // Six slices (VBX_FF_SLICES overrides, 1..8).  With k equal slices the call ends about one slice's scan after the last
    // resonance exists, so more slices shorten the exposed tail -- until a slice no longer fills the GPU: the root finder
    // is a chain of dependent operations per lane and a launch takes one wavefront's run time (0.47 ms at order 12) however
    // few wavefronts it has.  Measured, frames/s at 1 M x 512 for k = 4, 5, 6, 7, 8: 140, 144, 153, 142, 136 M; k = 6 is also
    // the best or within 1 % of it at 0.3, 0.7, 1.3 and 2 M frames and in the 4.5 M-frame pipeline.
    static const int want_slices = [] { const char *e = getenv("VBX_FF_SLICES"); const int v = e ? atoi(e) : 6; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
    const int n_slices = seg_len > 0 ? want_slices : 1;
    const long tc = (seg_len + n_slices - 1) / n_slices;
    ctx->last_track.res = res; ctx->last_track.cnt = cnt; ctx->last_track.st = st; ctx->last_track.F = F;
    ctx->last_track.n_est = (int)n_est; ctx->last_track.out = (res_t *)out_formants; ctx->last_track.out_ld = (long)formants_ld;
    if (n_slices == 1) {
        rc = run_burg(ctx, stm, x, pcm, F, (int)frame_len, (long)stride, hann, p, coeffs, st);                                      // :75
        if (rc != VBX_SUCCESS) return rc;
        rc = run_formant_resonances(ctx, stm, coeffs, F, p, sample_rate, res, cnt, st);                                            // :80-110
        if (rc != VBX_SUCCESS) return rc;
        rc = run_tracker(ctx, stm, chunked, res, F, VBX_MAX_RESONANCES, cnt, d_seg, (long)nseg, d_est, (int)n_est, st,
                         (res_t *)out_formants, (long)formants_ld);                                                                    // :114
        if (rc != VBX_SUCCESS) return rc;
        return check_launch(ctx, "vbx_find_formants_f64");
    }
    if (!ctx->trk) {
        VBX_HIP(ctx, hipStreamCreateWithFlags(&ctx->trk, hipStreamNonBlocking));
        for (auto &e : ctx->ev_slice) VBX_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        VBX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_trk, hipEventDisableTiming));
    }
    for (int j = 0; j < n_slices && j * tc < seg_len; j++) {
        const frame_map_t map{seg_len, j * tc, (seg_len - j * tc < tc) ? seg_len - j * tc : tc};   // the last slice may be shorter
        rc = run_burg(ctx, stm, x, pcm, F, (int)frame_len, (long)stride, hann, p, coeffs, st, map);
        if (rc != VBX_SUCCESS) return rc;
        rc = run_formant_resonances(ctx, stm, coeffs, F, p, sample_rate, res, cnt, st, map);
        if (rc != VBX_SUCCESS) return rc;
        VBX_HIP(ctx, hipEventRecord(ctx->ev_slice[j], stm));
        VBX_HIP(ctx, hipStreamWaitEvent(ctx->trk, ctx->ev_slice[j], 0));
        { Prof pr(ctx, "tracker", ctx->trk); launch_tracker(ctx->trk, res, F, VBX_MAX_RESONANCES, cnt, d_seg, (long)nseg, d_est, (int)n_est, st, (res_t *)out_formants, (long)formants_ld, j * tc, tc); }
    }
    VBX_HIP(ctx, hipEventRecord(ctx->ev_trk, ctx->trk));
    VBX_HIP(ctx, hipStreamWaitEvent(stm, ctx->ev_trk, 0));                // join: the formant tracks are complete on stm
    return check_launch(ctx, "vbx_find_formants_f64");
}

int vbx_find_formants_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len,
                          size_t stride, double sample_rate, size_t n_coeffs,
                          const int64_t *h_seg_start, size_t n_segments,
                          const vbx_resonance *h_est_init, size_t n_est,
                          vbx_resonance *out_formants, vbx_resonance *out_res, int32_t *out_res_count,
                          double *out_coeffs, int32_t *status) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride, VBX_MAX_LONG_FRAME_LEN);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    return run_find_formants(ctx, ctx->stream, x, n_frames, frame_len, stride, sample_rate, n_coeffs, h_seg_start, n_segments,
                             h_est_init, n_est, out_formants, 2 * n_est, out_res, out_res_count, out_coeffs, status);
}

// The tracks of the LAST vbx_find_formants_f64 / vbx_analyze_frames_* call on this context, continued from the true state
// before frame `first` (k_tracker.hip, tracker_stitch_kernel); stream: the context's, or the communicator's (vbx_comm.hip)
int vbx_internal_track_stitch(vbx_ctx *ctx, void *stream, vbx_resonance *formants, size_t n_frames, size_t formants_ld,
                              size_t first, size_t stop, const double *d_state_in, int32_t *d_changed) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    VBX_REQUIRE(ctx, formants && d_state_in, "null argument");
    const auto &lt = ctx->last_track;
    VBX_REQUIRE(ctx, lt.res != nullptr && lt.out == (res_t *)formants && lt.F == (long)n_frames && lt.out_ld == (long)formants_ld,
                "the formant rows are not the ones the last find_formants / analyze_frames call on this context wrote");
    VBX_REQUIRE(ctx, first <= stop && stop <= n_frames, "need first <= stop <= n_frames");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    { Prof p(ctx, "tracker_stitch", st);
      launch_tracker_stitch(st, lt.res, lt.F, VBX_MAX_RESONANCES, lt.cnt, lt.n_est, lt.st, lt.out, lt.out_ld, (long)first, (long)stop,
                            d_state_in, d_changed); }
    return check_launch(ctx, "vbx_track_stitch_f64");
}

int vbx_track_stitch_f64(vbx_ctx *ctx, vbx_resonance *formants, size_t n_frames, size_t formants_ld, size_t first, size_t stop,
                         const vbx_resonance *d_state_in, int32_t *d_changed) {
    return vbx_internal_track_stitch(ctx, nullptr, formants, n_frames, formants_ld, first, stop, (const double *)d_state_in, d_changed);
}

// 2 * VBX_FORMANT_SLOTS doubles of context-owned device memory (the state a communicator receives from the previous rank)
double *vbx_internal_stitch_state(vbx_ctx *ctx) {
    if (!ctx) return nullptr;
    if (!ctx->stitch_state) {
        if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
        if (hipMalloc((void **)&ctx->stitch_state, (2 * VBX_FORMANT_SLOTS + 2) * sizeof(double)) != hipSuccess) { ctx->stitch_state = nullptr; return nullptr; }
    }
    return ctx->stitch_state;
}
int vbx_internal_last_track_n_est(vbx_ctx *ctx) { return ctx ? ctx->last_track.n_est : 0; }
int vbx_internal_last_spectral_split(vbx_ctx *ctx) { return ctx ? ctx->last_spectral_split : 0; }
int vbx_internal_last_mfcc_interp(vbx_ctx *ctx) { return ctx ? ctx->last_mfcc_interp : 0; }
// every host-side condition of a stitch / hand-off on these rows, for callers that must know BEFORE they enqueue anything
// a peer waits for (vbx_comm.hip: an early return between ncclRecv and ncclSend would leave the next rank blocked)
int vbx_internal_track_check(vbx_ctx *ctx, const vbx_resonance *formants, size_t n_frames, size_t formants_ld) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    const auto &lt = ctx->last_track;
    VBX_REQUIRE(ctx, lt.res != nullptr && lt.n_est >= 1, "no tracks on this context: the last call produced none");
    VBX_REQUIRE(ctx, formants && lt.out == (const res_t *)formants && lt.F == (long)n_frames && lt.out_ld == (long)formants_ld,
                "the formant rows are not the ones the last find_formants / analyze_frames call on this context wrote");
    return VBX_SUCCESS;
}

// Host only (no device, no context): the tables of the MFCC bins interpolated inside the fused kernel (mfcc_interp_t) for one shape,
// for tests that hold the interpolation to the frame's exact DFT.  desc[8] = {M, threads per frame, slots, taps, jmin, jmax,
// byte offset of the taps, byte offset of the first-tap indices}; *need = bytes of the table (copied to buf when cap >= *need).
// Returns 1 when the shape has the form, 0 when it has not (its MFCC comes from the chirp-z kernel), < 0 on a bad argument.
int vbx_internal_mfcc_interp_table(size_t frame_len, int b_lo, int nb, int32_t *desc, void *buf, size_t cap, size_t *need) {
    if (!desc || !need || frame_len < 2 || frame_len > VBX_MAX_FRAME_LEN || nb < 1 || nb > 4096) return VBX_E_INVALID;
    const int plan = spectral_plan_mfcc((int)frame_len);                  // (the plan vbx_analyze_frames_f64 picks)
    if (plan == SPECTRAL_PLAN_NONE || (2 * spectral_plan_nc(plan)) % (int)frame_len == 0) return 0;
    const size_t bytes = mfcc_interp_table_bytes(plan, nb);
    *need = bytes;
    std::vector<char> h(bytes, 0);
    mfcc_interp_t d{};
    if (!mfcc_interp_fill(plan, (int)frame_len, b_lo, nb, h.data(), &d)) return 0;
    const int nt = plan == SPECTRAL_PLAN_4096 ? 128 : 64;
    desc[0] = 4 * spectral_plan_nc(plan) / 2; desc[1] = nt; desc[2] = (nb + nt - 1) / nt; desc[3] = d.taps;
    desc[4] = d.jmin; desc[5] = d.jmax; desc[6] = (int32_t)mfcc_interp_coef_offset(plan); desc[7] = (int32_t)mfcc_interp_j0_offset(plan, nb);
    if (buf && cap >= bytes) std::memcpy(buf, h.data(), bytes);
    return 1;
}

// ---- spectrum.rs: MFCC --------------------------------------------------------------------

static int run_mfcc(vbx_ctx *ctx, hipStream_t stm, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                    const double *window, size_t num_coeffs, double lo_hz, double hi_hz,
                    double sample_rate, double *out, size_t out_ld, int32_t *status) {
    VBX_REQUIRE(ctx, out != nullptr, "null output");
    VBX_REQUIRE(ctx, num_coeffs >= 1 && num_coeffs <= 64, "num_coeffs must be in [1, 64]");
    VBX_REQUIRE(ctx, out_ld >= num_coeffs, "output rows must hold num_coeffs entries");
    std::vector<int32_t> hb; bool bad = false; const int32_t *d_bins = nullptr;
    ctx->last_mfcc_interp = 0;
    int rc = get_bins_dev(ctx, frame_len, num_coeffs, lo_hz, hi_hz, sample_rate, &d_bins, hb, bad);
    if (rc != VBX_SUCCESS) return rc;
    if (bad) {   // the reference panics on every frame (bins do not depend on the data)
        { Prof p(ctx, "fill_rows", stm); launch_fill_rows(stm, out, (long)n_frames, (int)num_coeffs, (long)out_ld, 0.0, status, VBX_FRAME_ERR_PANIC); }
        return check_launch(ctx, "vbx_mfcc_f64");
    }
    const int nb = hb.back() - hb.front();
    const double *dct = nullptr, *slopes = nullptr;
    rc = get_dct_dev(ctx, num_coeffs, &dct); if (rc != VBX_SUCCESS) return rc;
    rc = get_slopes_dev(ctx, frame_len, num_coeffs, lo_hz, hi_hz, sample_rate, hb, &slopes); if (rc != VBX_SUCCESS) return rc;
    if (frame_len > VBX_MAX_FRAME_LEN) {          // a long frame: the Goertzel recurrence over HBM, the filter sums' inputs in a scratch (k_long.hip)
        VBX_REQUIRE(ctx, nb >= 1, "no mel bins");
        const double *tw = nullptr;
        rc = get_goertzel_dev(ctx, frame_len, hb.front(), nb, &tw); if (rc != VBX_SUCCESS) return rc;
        void *w = nullptr;
        rc = ws_get(ctx, vbx_ctx::WS_CZT, mfcc_long_scratch_bytes((long)n_frames, nb), &w); if (rc != VBX_SUCCESS) return rc;
        { Prof p(ctx, "mfcc_long", stm);
          launch_mfcc_long(stm, x, (long)n_frames, (long)frame_len, (long)stride, window, tw, d_bins, slopes, dct, (int)num_coeffs, nb, out,
                           (long)out_ld, status, (double *)w); }
        return check_launch(ctx, "vbx_mfcc_f64");
    }
    // frames that fill one of the FFT kernels' transforms (1024, 1200, 2048, 4096): the forward half of the fused spectral
    // kernel -- one real FFT of the zero-padded frame, whose even bins are the n-point DFT the mel filters read
    {
        int plan = spectral_plan((int)frame_len);
        // 2048 and 4096 samples: the frame as the real sequence of the half-size transform (complex 1024 / 2048), no padding
        if (frame_len == 2048) plan = SPECTRAL_PLAN_1024; else if (frame_len == 4096) plan = SPECTRAL_PLAN_2048;
        if (!ctx->mfcc_force_goertzel && !ctx->mfcc_force_dft2 && !ctx->mfcc_force_mfma && plan != SPECTRAL_PLAN_NONE &&
            ((int)frame_len == spectral_plan_nc(plan) || (int)frame_len == 2 * spectral_plan_nc(plan)) &&
            num_coeffs <= 64 && nb >= 1 && hb.front() >= 0 && hb.front() + nb <= (int)frame_len / 2) {
            const double *tab = nullptr;
            rc = get_spectral_tab(ctx, plan, &tab); if (rc != VBX_SUCCESS) return rc;
            spectral_launch_t L{};
            L.plan = plan; L.n = (int)frame_len; L.mfcc_only = true;
            L.x = x; L.F = (long)n_frames; L.stride = (long)stride; L.window = window; L.tab = tab;
            L.out_mfcc = out; L.mfcc_ld = (long)out_ld; L.mfcc_status = status;
            L.bins = d_bins; L.slopes = slopes; L.dct = dct; L.num_coeffs = (int)num_coeffs; L.nb = nb;
            { Prof p(ctx, "mfcc", stm); launch_analyze(stm, L); }
            return check_launch(ctx, "vbx_mfcc_f64");
        }
    }
    // composite frame lengths: two-stage DFT of the needed bins, on the matrix cores when the factorisation fits
    // the MFMA kernel's tiles, else on the vector ALU; otherwise (prime-ish lengths) Goertzel.  Every kernel writes
    // status 0 itself (no separate memset queued behind whatever the stream is running).
    const bool composite_ok = nb > 0 && !ctx->mfcc_force_goertzel;
    const mfcc_mplan_t mp = (composite_ok && !ctx->mfcc_force_dft2 && ctx->mfcc_czt != 1) ? mfcc_mfma_plan((int)frame_len, hb.front(), nb) : mfcc_mplan_t{};
    const mfcc_plan_t pl = (composite_ok && !mp.ok) ? mfcc_plan((int)frame_len, nb) : mfcc_plan_t{false, 0, 0, 0, 0};
    // Round 5: the forward transform of the zero-padded frame that the fused kernels use, the frame's DFT bins interpolated from the
    // transform's (mfcc_interp_t, vbx_kernels.hpp: 24-40 taps per bin, error < 1e-14 of the largest bin) -- one transform instead of the
    // chirp-z kernel's two.  Where the matrix-core kernel has no plan, and from 1400 samples up where it has (measured, M frames/s,
    // before -> interpolated: 1103: 63 -> 110, 2047: 30 -> 54, 3000: 29 -> 34, 4000: 12 -> 26; 1500 / 1600 / 1800: 61 -> 71, 62 -> 73,
    // 48 -> 67; the matrix-core kernel stays at 700 / 882 / 1280: 209 / 161 / 94 against 133 / 123 / 74).
    // VBX_MFCC_INTERP=0 / the force switches: the kernels below, as before.
    if ((!mp.ok || frame_len >= 1400) && ctx->mfcc_interp != 0 && ctx->mfcc_czt != 1 && !ctx->mfcc_force_goertzel && !ctx->mfcc_force_dft2 && !ctx->mfcc_force_mfma &&
        num_coeffs <= 64 && nb >= 1 && nb <= 4096 && hb.front() >= 0 && hb.front() + nb <= (int)frame_len / 2) {
        const int plan = spectral_plan_mfcc((int)frame_len);
        bool ok = false;
        mfcc_interp_t ip{};
        if (plan != SPECTRAL_PLAN_NONE && (2 * spectral_plan_nc(plan)) % (int)frame_len != 0 && (int)frame_len < spectral_plan_nc(plan)) {
            rc = get_interp_dev(ctx, plan, (int)frame_len, hb.front(), nb, &ip, &ok); if (rc != VBX_SUCCESS) return rc;
        }
        if (ok) {
            const double *tab = nullptr;
            rc = get_spectral_tab(ctx, plan, &tab); if (rc != VBX_SUCCESS) return rc;
            spectral_launch_t L{};
            L.plan = plan; L.n = (int)frame_len; L.mfcc_only = true; L.interp = true; L.ip = ip;
            ctx->last_mfcc_interp = 1;
            L.x = x; L.F = (long)n_frames; L.stride = (long)stride; L.window = window; L.tab = tab;
            L.out_mfcc = out; L.mfcc_ld = (long)out_ld; L.mfcc_status = status;
            L.bins = d_bins; L.slopes = slopes; L.dct = dct; L.num_coeffs = (int)num_coeffs; L.nb = nb;
            { Prof p(ctx, "mfcc", stm); launch_analyze(stm, L); }
            return check_launch(ctx, "vbx_mfcc_f64");
        }
    }
    // no matrix-core factorisation (prime-ish lengths such as 1103 = 25 ms at 44.1 kHz, or too many bins for the two-stage
    // kernel's tiles: 2500, 3000): the needed bins by the chirp-z identity on a power-of-two transform (two complex FFTs per
    // frame, k_mfcc_czt.hip) instead of evaluating them bin by bin on the vector ALU.  Measured, MFCC alone / the whole
    // pipeline, M frames/s: 1103: 28.9 -> 61.9 / 15.2 -> 20.3; 2500: 7.4 -> 15.0 / 3.2 -> 4.0; 3000: 2.8 -> 15.2 / 1.8 -> 3.9.
    // Where the matrix-core kernel has a plan it stays (1000, 1102, 1800: it is the faster one; 1500, 1600: within 5 %).
    const int czt_top = hb.back();
    int czt_plan = (nb >= 1 && hb.front() >= 0 && num_coeffs <= 64) ? mfcc_czt_plan((int)frame_len, czt_top) : SPECTRAL_PLAN_NONE;
    // too long for one transform (frame_len + top - 1 > 4096: 3,431..4,095 samples at these settings), or VBX_MFCC_CZT_SPLIT=1
    // (tests): the frame in two halves, each with its own chirp segment, the complex results summed (vbx_mfcc_czt.hpp) -- four
    // 4096-point transforms per frame instead of frame_len x bins products (pipeline at 4000 / 2000: 1.4 -> 5.7 M frames/s)
    int czt_n1 = 0;
    if (nb >= 1 && hb.front() >= 0 && num_coeffs <= 64 && (czt_plan == SPECTRAL_PLAN_NONE || ctx->mfcc_czt_split)) {
        int n1 = 0;
        const int p2 = mfcc_czt_split_plan((int)frame_len, czt_top, &n1);
        if (p2 != SPECTRAL_PLAN_NONE) { czt_plan = p2; czt_n1 = n1; }
    }
    // ... and from 1400 samples up to what the 2048-point transform holds, where the matrix-core kernel HAS a plan: inside the
    // pipeline its 512-thread workgroups with ~100 KB of LDS keep the analyze kernel's wavefronts out, the chirp-z kernel's
    // one-wavefront workgroups interleave with them (pipeline at 1500 / 1600 samples: 15.6 -> 16.2, 15.1 -> 16.8 M frames/s;
    // below 1400 and on the 4096-point transform the matrix-core kernel wins: 1280: 19.6 against 17.6, 1800: 14.8 against 12.5)
    const bool czt_over_mfma = mp.ok && czt_plan == SPECTRAL_PLAN_2048 && czt_n1 == 0 && frame_len >= 1400 && ctx->mfcc_czt == -1 &&
                               !(ctx->mfcc_force_goertzel || ctx->mfcc_force_dft2 || ctx->mfcc_force_mfma);
    if (!mp.ok || czt_over_mfma) {
        const int top = czt_top;
        const int cplan = czt_plan;
        const bool forced = ctx->mfcc_force_goertzel || ctx->mfcc_force_dft2 || ctx->mfcc_force_mfma;
        // (below ~600 samples the Goertzel kernel's n * nb products cost less than two 1024-point transforms)
        const bool want = ctx->mfcc_czt == 1 || czt_over_mfma || (ctx->mfcc_czt == -1 && !forced && frame_len >= 600);
        if (cplan != SPECTRAL_PLAN_NONE && want) {
            const double *tab = nullptr, *chirp = nullptr, *bhat = nullptr;
            rc = get_spectral_tab(ctx, cplan, &tab); if (rc != VBX_SUCCESS) return rc;
            rc = get_czt_dev(ctx, frame_len, top, spectral_plan_nc(cplan), czt_n1, &chirp, &bhat); if (rc != VBX_SUCCESS) return rc;
            void *cw = nullptr;
            rc = ws_get(ctx, vbx_ctx::WS_CZT, (czt_n1 ? 4 : 2) * (size_t)spectral_plan_nc(cplan) * sizeof(double), &cw); if (rc != VBX_SUCCESS) return rc;
            { Prof p(ctx, "mfcc", stm);
              launch_mfcc_czt(stm, cplan, x, (long)n_frames, (int)frame_len, czt_n1, (long)stride, window, tab, chirp, bhat, d_bins, slopes, dct,
                              (int)num_coeffs, nb, out, (long)out_ld, status, (double *)cw); }
            return check_launch(ctx, "vbx_mfcc_f64");
        }
    }
    if (mp.ok) {
        const double *ctab = nullptr, *twd = nullptr, *twm = nullptr, *wm = nullptr;
        rc = get_mfcc_mfma_dev(ctx, frame_len, mp, &ctab, &twd, &twm, &wm); if (rc != VBX_SUCCESS) return rc;
        Prof p(ctx, "mfcc", stm);
        launch_mfcc_mfma(stm, x, (long)n_frames, (int)frame_len, (long)stride, window, mp, ctab, twd, twm, wm, d_bins,
                         slopes, dct, (int)num_coeffs, out, (long)out_ld, status, nb, ctx->cu_count);
    } else if (pl.ok) {
        const double *ctab = nullptr, *twid = nullptr;
        rc = get_dft2_dev(ctx, frame_len, pl, &ctab, &twid); if (rc != VBX_SUCCESS) return rc;
        Prof p(ctx, "mfcc", stm);
        launch_mfcc_dft2(stm, x, (long)n_frames, (int)frame_len, (long)stride, window, pl, ctab, twid, d_bins,
                         slopes, dct, (int)num_coeffs, out, (long)out_ld, status, nb, ctx->cu_count);
    } else {
        VBX_REQUIRE(ctx, mfcc_fits((int)frame_len, nb), "frame / bin range does not fit the LDS");
        const double *tw = nullptr;
        rc = get_goertzel_dev(ctx, frame_len, hb.front(), nb, &tw); if (rc != VBX_SUCCESS) return rc;
        Prof p(ctx, "mfcc", stm);
        launch_mfcc(stm, x, (long)n_frames, (int)frame_len, (long)stride, window, tw, d_bins, slopes, dct, (int)num_coeffs, out, (long)out_ld, status, nb);
    }
    return check_launch(ctx, "vbx_mfcc_f64");
}

int vbx_mfcc_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                 const double *window, size_t num_coeffs, double lo_hz, double hi_hz,
                 double sample_rate, double *out, int32_t *status) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride, VBX_MAX_LONG_FRAME_LEN);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    return run_mfcc(ctx, ctx->stream, x, n_frames, frame_len, stride, window, num_coeffs, lo_hz, hi_hz, sample_rate,
                    out, num_coeffs, status);
}

int vbx_dct_f64(vbx_ctx *ctx, const double *in, size_t n_rows, size_t n, double *out) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_rows == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, in && out && n >= 1 && n <= 4096 && n_rows <= 0x7fffffffull, "bad argument");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    const double *dct = nullptr;
    int rc = get_dct_dev(ctx, n, &dct); if (rc != VBX_SUCCESS) return rc;
    { Prof p(ctx, "dct_rows"); launch_dct_rows(ctx->stream, in, (long)n_rows, (int)n, dct, out); }
    return check_launch(ctx, __func__);
}

// ---- front end (N2, N3) ---------------------------------------------------------------------

int vbx_pcm16_to_f64(vbx_ctx *ctx, const int16_t *pcm, size_t n_samples, double *out) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_samples == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, pcm && out, "null argument");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "pcm16"); launch_pcm16(ctx->stream, pcm, n_samples, 32767.0, out); }
    return check_launch(ctx, __func__);
}

int vbx_rms_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                const double *window, double *out) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride, VBX_MAX_LONG_FRAME_LEN);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr, "null output");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "rms"); launch_rms(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, window, out); }
    return check_launch(ctx, __func__);
}

int vbx_preemphasis_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                        double factor, double *out) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride, VBX_MAX_LONG_FRAME_LEN);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr, "null output");
    VBX_REQUIRE(ctx, out != x || stride == frame_len, "in-place filtering needs a dense batch (stride == frame_len)");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    if (frame_len > VBX_MAX_FRAME_LEN) {          // a whole signal (src/waves.rs:86 takes any slice): tiles + a carry pass (k_long.hip)
        void *w = nullptr;
        rc = ws_get(ctx, vbx_ctx::WS_LONG, preemphasis_long_scratch_bytes((long)n_frames, (long)frame_len), &w);
        if (rc != VBX_SUCCESS) return rc;
        { Prof p(ctx, "preemphasis_long"); launch_preemphasis_long(ctx->stream, x, (long)n_frames, (long)frame_len, (long)stride, 2.0 * M_PI * factor, out, (double *)w); }
        return check_launch(ctx, __func__);
    }
    { Prof p(ctx, "preemphasis"); launch_preemphasis(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, 2.0 * M_PI * factor, out); }
    return check_launch(ctx, __func__);
}

int vbx_ring_frames_f64(vbx_ctx *ctx, const double *ring, size_t capacity, size_t head, size_t n_frames,
                        size_t frame_len, size_t stride, double *out) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_frames == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, ring && out, "null argument");
    VBX_REQUIRE(ctx, capacity >= 1 && head < capacity, "head must lie inside the ring");
    VBX_REQUIRE(ctx, frame_len >= 1 && frame_len <= VBX_MAX_LONG_FRAME_LEN && stride >= 1, "bad frame geometry");
    VBX_REQUIRE(ctx, (n_frames - 1) * stride + frame_len <= capacity, "the view is longer than the deque can be");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "ring_frames"); launch_ring_frames(ctx->stream, ring, (long)capacity, (long)head, (long)n_frames, (int)frame_len, (long)stride, out); }
    return check_launch(ctx, __func__);
}

int vbx_resample_linear_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                            double resample_ratio, double *out) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride, VBX_MAX_LONG_FRAME_LEN);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr, "null output");
    VBX_REQUIRE(ctx, resample_ratio > 0.0 && resample_ratio <= 64.0, "resample_ratio must be in (0, 64]");
    const size_t m = vbx_resampled_len(frame_len, resample_ratio);
    VBX_REQUIRE(ctx, m >= 1 && m <= 0x3fffffff, "bad resampled length");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    auto key = std::make_pair(frame_len, resample_ratio);
    auto it = ctx->resample_tabs.find(key);
    if (it == ctx->resample_tabs.end()) {
        // sample 0.10 Converter: interpolation_value starts at 0, grows by 1/ratio per output, and every whole
        // unit advances the (left, right) pair by one source sample; left starts at source index 0
        std::vector<int32_t> hi(m);
        std::vector<double> hf(m);
        double value = 0.0;
        const double step = 1.0 / resample_ratio;
        long left = 0;
        for (size_t k = 0; k < m; k++) {
            while (value >= 1.0) { left++; value -= 1.0; }
            hi[k] = (left > 0x3fffffff) ? 0x3fffffff : (int32_t)left;
            hf[k] = value;
            value += step;
        }
        int32_t *di = nullptr; double *df = nullptr;
        VBX_HIP(ctx, hipMalloc((void **)&di, m * sizeof(int32_t)));
        VBX_HIP(ctx, hipMalloc((void **)&df, m * sizeof(double)));
        VBX_HIP(ctx, hipMemcpy(di, hi.data(), m * sizeof(int32_t), hipMemcpyHostToDevice));
        VBX_HIP(ctx, hipMemcpy(df, hf.data(), m * sizeof(double), hipMemcpyHostToDevice));
        it = ctx->resample_tabs.emplace(key, std::make_pair(di, df)).first;
    }
    { Prof p(ctx, "resample"); launch_resample(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, it->second.first, it->second.second, (int)m, out); }
    return check_launch(ctx, __func__);
}

// ---- the user's frame loop, batched and fused ------------------------------------------------------------------

static int ensure_side_stream(vbx_ctx *ctx) {
    if (ctx->side) return VBX_SUCCESS;
    VBX_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    VBX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    VBX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    return VBX_SUCCESS;
}

// x: the frames as f64 samples, or -- pcm16 non-null -- as 16-bit PCM (the kernels that have a PCM form read it directly:
// 1200-sample frames through the fused spectral kernel, Burg at every length; every other shape is widened into a
// context-owned f64 copy of the view first and takes the f64 path)
static int analyze_frames_impl(vbx_ctx *ctx, const char *fn, const double *x, const int16_t *pcm16, size_t n_frames, size_t frame_len,
                               size_t stride, const vbx_analysis_params *h_p, const int64_t *h_seg_start, size_t n_segments,
                               double *out_records, size_t record_ld, int32_t *status3) {
    int rc = check_frames(ctx, fn, pcm16 ? (const void *)pcm16 : (const void *)x, n_frames, frame_len, stride, VBX_MAX_LONG_FRAME_LEN);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, h_p && out_records, "null argument");
    const size_t rec = vbx_record_doubles(h_p);
    VBX_REQUIRE(ctx, record_ld >= rec && record_ld % 2 == 0, "record_ld must be even and >= vbx_record_doubles(params)");
    VBX_REQUIRE(ctx, ((uintptr_t)out_records & 15) == 0, "records must be 16-byte aligned");
    VBX_REQUIRE(ctx, !h_p->formant_order || (h_p->n_est >= 1 && h_p->n_est <= VBX_FORMANT_SLOTS), "n_est must be in [1, 6]");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    rc = ensure_side_stream(ctx);
    if (rc != VBX_SUCCESS) return rc;
    // One spectral pass for pitch + LPC + MFCC when the shape has a fused kernel (k_spectral.hip); otherwise the LPC
    // and MFCC kernels run on the side stream and the pitch kernel alone on the main one.
    std::vector<int32_t> hb; bool bad_bins = false; const int32_t *d_bins = nullptr;
    const double *dct = nullptr, *slopes = nullptr;
    int nb = 0;
    if (h_p->mfcc_coeffs) {
        VBX_REQUIRE(ctx, h_p->mfcc_coeffs <= 64, "mfcc_coeffs must be in [0, 64]");
        rc = get_bins_dev(ctx, frame_len, h_p->mfcc_coeffs, h_p->mfcc_lo_hz, h_p->mfcc_hi_hz, h_p->sample_rate, &d_bins, hb, bad_bins);
        if (rc != VBX_SUCCESS) return rc;
        nb = hb.back() - hb.front();
    }
    // (MFCC joins the fused kernel when the frame's length divides the transform's: n = 512, 600, 800, 1024, 1200, 2048, 4096 -- and, since
    // round 5, at every other length by interpolated bins, below; LPC only at the order the kernel's register Levinson is built for -- what
    // cannot join runs from its own kernel on the side stream)
    const bool fused = !ctx->pitch_force_mfma && spectral_supported((int)frame_len, 0, 0, 0, 0);
    const bool fused_lpc = fused && h_p->lpc_order == SPECTRAL_LPC_ORDER;
    // the transform: the one its length asks for, or -- if MFCC can join only there -- the one whose length the frame divides
    int plan = fused ? spectral_plan((int)frame_len) : SPECTRAL_PLAN_NONE;
    if (fused && !bad_bins && h_p->mfcc_coeffs) {
        const int pm = spectral_plan_mfcc((int)frame_len);
        if (spectral_supported_plan(pm, (int)frame_len, 0, nb, hb.front(), (int)h_p->mfcc_coeffs)) plan = pm;
    }
    bool fused_mfcc = fused && !bad_bins && h_p->mfcc_coeffs &&
                      spectral_supported_plan(plan, (int)frame_len, 0, nb, hb.front(), (int)h_p->mfcc_coeffs);
    // ... and at the other lengths by interpolating the frame's DFT bins from the transform's (mfcc_interp_t)
    bool interp_mfcc = false;
    mfcc_interp_t ip{};
    if (fused && !fused_mfcc && !bad_bins && h_p->mfcc_coeffs && ctx->mfcc_interp != 0 && plan != SPECTRAL_PLAN_NONE && nb >= 1 && nb <= 4096 && hb.front() >= 0) {
        rc = get_interp_dev(ctx, plan, (int)frame_len, hb.front(), nb, &ip, &interp_mfcc);
        if (rc != VBX_SUCCESS) return rc;
        fused_mfcc = interp_mfcc;
    }
    // 16-bit PCM frames: the fused kernel of full 1200-sample frames, the pitch fallback and Burg read them directly;
    // anything that would send another kernel over the samples takes one widening pass into a context-owned f64 copy
    const bool pcm_native = pcm16 != nullptr && fused && frame_len == (size_t)SPECTRAL_N &&
                            (!h_p->lpc_order || fused_lpc) && (!h_p->mfcc_coeffs || fused_mfcc);
    if (pcm16 != nullptr && !pcm_native) {
        const size_t ns = (n_frames - 1) * stride + frame_len;
        void *w = nullptr;
        rc = ws_get(ctx, vbx_ctx::WS_F32_IN, ns * sizeof(double), &w);
        if (rc != VBX_SUCCESS) return rc;
        { Prof p(ctx, "pcm16"); launch_pcm16(ctx->stream, pcm16, ns, 32767.0, (double *)w); }
        x = (const double *)w; pcm16 = nullptr;
    }
    if (pcm_native) x = reinterpret_cast<const double *>(pcm16);          // the PCM kernels take the pointer through the f64 slot
    const double *hann = nullptr;
    rc = get_window_dev(ctx, VBX_WINDOW_HANNING, frame_len, &hann);        // Windower::hanning frames (examples/pitch_detection.rs:23)
    if (rc != VBX_SUCCESS) return rc;
    // record columns
    const size_t c_form = 2, c_mfcc = c_form + (h_p->formant_order ? 2 * h_p->n_est : 0),
                 c_lpc = c_mfcc + h_p->mfcc_coeffs;
    int32_t *st_pitch = nullptr, *st_form = nullptr, *st_mfcc = nullptr;
    if (status3) { st_pitch = status3; st_form = status3 + n_frames; st_mfcc = status3 + 2 * n_frames; }
    // fork: the formant chain (Burg -> roots -> the latency-bound tracker scan) and the MFCC run on the side stream,
    // beside the FP64-bound pitch kernel
    VBX_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    // (frames longer than VBX_MAX_FRAME_LEN: everything in order on the context's stream -- the long-frame kernels share one scratch)
    hipStream_t side = frame_len > VBX_MAX_FRAME_LEN ? ctx->stream : ctx->side;
    VBX_HIP(ctx, hipStreamWaitEvent(side, ctx->ev_fork, 0));
    if (h_p->formant_order) {
        vbx_resonance est[VBX_FORMANT_SLOTS];
        for (size_t e = 0; e < h_p->n_est; e++) est[e] = h_p->est_init[e];
        rc = run_find_formants(ctx, side, x, n_frames, frame_len, stride, h_p->sample_rate, h_p->formant_order,
                               h_seg_start, n_segments, est, h_p->n_est, (vbx_resonance *)(out_records + c_form), record_ld,
                               nullptr, nullptr, nullptr, st_form, pcm_native ? pcm16 : nullptr);
        if (rc != VBX_SUCCESS) return rc;
    } else {
        ctx->last_track.res = nullptr;                        // no tracks in these records: nothing for vbx_track_stitch_f64 to continue
        ctx->last_track.n_est = 0;                            // ... and no row for a communicator to send on
        if (st_form) VBX_HIP(ctx, hipMemsetAsync(st_form, 0, n_frames * sizeof(int32_t), side));
    }
    if (fused && h_p->lpc_order && !fused_lpc) {
        rc = run_autocorr_lpc(ctx, side, x, n_frames, frame_len, stride, hann, h_p->lpc_order, 0, nullptr,
                              out_records + c_lpc, record_ld);
        if (rc != VBX_SUCCESS) return rc;
    }
    if (fused && h_p->mfcc_coeffs && !fused_mfcc) {
        rc = run_mfcc(ctx, side, x, n_frames, frame_len, stride, hann, h_p->mfcc_coeffs, h_p->mfcc_lo_hz, h_p->mfcc_hi_hz,
                      h_p->sample_rate, out_records + c_mfcc, record_ld, st_mfcc);
        if (rc != VBX_SUCCESS) return rc;
    }
    if (!fused) {
        if (h_p->lpc_order) {
            rc = run_autocorr_lpc(ctx, side, x, n_frames, frame_len, stride, hann, h_p->lpc_order, 0, nullptr,
                                  out_records + c_lpc, record_ld);
            if (rc != VBX_SUCCESS) return rc;
        }
        if (h_p->mfcc_coeffs) {
            rc = run_mfcc(ctx, side, x, n_frames, frame_len, stride, hann, h_p->mfcc_coeffs, h_p->mfcc_lo_hz, h_p->mfcc_hi_hz,
                          h_p->sample_rate, out_records + c_mfcc, record_ld, st_mfcc);
            if (rc != VBX_SUCCESS) return rc;
        }
    }
    if (!h_p->mfcc_coeffs && st_mfcc) VBX_HIP(ctx, hipMemsetAsync(st_mfcc, 0, n_frames * sizeof(int32_t), side));
    VBX_HIP(ctx, hipEventRecord(ctx->ev_join, side));
    if (fused) {
        const double *lagw = nullptr, *tab = nullptr;
        rc = get_window_dev(ctx, VBX_WINDOW_HANNING_LAG, frame_len, &lagw); if (rc != VBX_SUCCESS) return rc;
        rc = get_spectral_tab(ctx, plan, &tab); if (rc != VBX_SUCCESS) return rc;
        if (fused_mfcc) {
            rc = get_dct_dev(ctx, h_p->mfcc_coeffs, &dct); if (rc != VBX_SUCCESS) return rc;
            rc = get_slopes_dev(ctx, frame_len, h_p->mfcc_coeffs, h_p->mfcc_lo_hz, h_p->mfcc_hi_hz, h_p->sample_rate, hb, &slopes);
            if (rc != VBX_SUCCESS) return rc;
        }
        if (ctx->prof && !ctx->pitch_work) {
            const size_t wb = PITCH_WORK_WORDS * sizeof(unsigned long long);
            VBX_HIP(ctx, hipMalloc((void **)&ctx->pitch_work, wb));
            VBX_HIP(ctx, hipMemsetAsync(ctx->pitch_work, 0, wb, ctx->stream));
        }
        spectral_launch_t L{};
        L.plan = plan; L.n = (int)frame_len;
        L.x = x; L.F = (long)n_frames; L.stride = (long)stride; L.window = hann; L.lag_window = lagw; L.tab = tab;
        L.lag_rcp = ctx->lag_rcp_ok.count(frame_len) && ctx->lag_rcp_ok[frame_len];
        L.sample_rate = h_p->sample_rate; L.threshold = h_p->pitch_threshold; L.fmin = h_p->pitch_fmin; L.fmax = h_p->pitch_fmax;
        L.kmax = 1;
        L.pcm = pcm_native;
        L.whole_curve = ctx->pitch_whole_curve;
        L.out_cand = (pitch_t *)out_records; L.cand_ld = (long)record_ld; L.out_count = nullptr; L.pitch_status = st_pitch;
        L.work = ctx->prof ? ctx->pitch_work : nullptr;
        if (fused_lpc) { L.out_lpc = out_records + c_lpc; L.lpc_ld = (long)record_ld; }
        if (fused_mfcc) {
            L.out_mfcc = out_records + c_mfcc; L.mfcc_ld = (long)record_ld; L.mfcc_status = st_mfcc;
            L.bins = d_bins; L.slopes = slopes; L.dct = dct; L.num_coeffs = (int)h_p->mfcc_coeffs; L.nb = nb;
            L.interp = interp_mfcc; L.ip = ip;
        }
        rc = launch_spectral(ctx, ctx->stream, L, "analyze");
    } else {
        rc = run_pitch(ctx, ctx->stream, x, n_frames, frame_len, stride, hann, h_p->sample_rate, h_p->pitch_threshold,
                       h_p->pitch_fmin, h_p->pitch_fmax, 1, (vbx_pitch *)out_records, record_ld, nullptr, st_pitch);
    }
    if (rc != VBX_SUCCESS) return rc;
    VBX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));       // join: the records are complete on ctx's stream
    return VBX_SUCCESS;
}

int vbx_analyze_frames_f64(vbx_ctx *ctx, const double *x, size_t n_frames, size_t frame_len, size_t stride,
                           const vbx_analysis_params *h_p, const int64_t *h_seg_start, size_t n_segments,
                           double *out_records, size_t record_ld, int32_t *status3) {
    return analyze_frames_impl(ctx, __func__, x, nullptr, n_frames, frame_len, stride, h_p, h_seg_start, n_segments, out_records,
                               record_ld, status3);
}

int vbx_analyze_frames_pcm16(vbx_ctx *ctx, const int16_t *pcm, size_t n_frames, size_t frame_len, size_t stride,
                             const vbx_analysis_params *h_p, const int64_t *h_seg_start, size_t n_segments,
                             double *out_records, size_t record_ld, int32_t *status3) {
    if (n_frames != 0 && ctx && !pcm) return fail(ctx, VBX_E_INVALID, "vbx_analyze_frames_pcm16: null frame pointer");
    return analyze_frames_impl(ctx, __func__, nullptr, pcm, n_frames, frame_len, stride, h_p, h_seg_start, n_segments, out_records,
                               record_ld, status3);
}

// ---- Sample = f32, the WIDE forms --------------------------------------------
// The traits are generic over the Sample type (src/periodic.rs:276-289 `T: Sample`, src/spectrum.rs:56 `T: Float`,
// :401-409).  The *_f32_wide entry points take float frames and return float results; samples are widened on load, the
// arithmetic runs in f64 and every result is rounded to f32 once -- more accurate than the reference's own f32 folds and as
// fast as the f64 kernels, but NOT the bits the crate returns at f32.  The reference-faithful forms (every fold in f32, in
// the reference's order) carry the plain *_f32 names, further down; MFCC has only the wide form (its f32 arithmetic lives
// in the un-vendored rustfft).

int vbx_autocorrelate_f32_wide(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                          size_t stride, const float *window, size_t n_lags, float *out) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr, "null output");
    VBX_REQUIRE(ctx, n_lags >= 1 && n_lags <= frame_len, "n_lags must be in [1, frame_len] (the reference panics beyond)");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    if (!fewlags_supported((int)frame_len, (int)n_lags, false) && !ctx->pitch_force_mfma &&
        spectral_plan((int)frame_len) != SPECTRAL_PLAN_NONE && (n_lags >= SPECTRAL_AC_MIN_LAGS || frame_len >= 1024)) {
        // many lags: widen (the windowed product rounded to f32 first), the f64 FFT path, one rounding to f32
        void *wi = nullptr, *wo = nullptr;
        rc = ws_get(ctx, vbx_ctx::WS_F32_IN, n_frames * frame_len * sizeof(double), &wi);
        if (rc != VBX_SUCCESS) return rc;
        rc = ws_get(ctx, vbx_ctx::WS_F32_OUT, n_frames * n_lags * sizeof(double), &wo);
        if (rc != VBX_SUCCESS) return rc;
        { Prof p(ctx, "widen_frames"); launch_widen_frames(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, window, (double *)wi); }
        rc = vbx_autocorrelate_f64(ctx, (const double *)wi, n_frames, frame_len, frame_len, nullptr, n_lags, (double *)wo);
        if (rc != VBX_SUCCESS) return rc;
        { Prof p(ctx, "narrow"); launch_narrow(ctx->stream, (const double *)wo, (long)(n_frames * n_lags), out); }
        return check_launch(ctx, __func__);
    }
    if (fewlags_supported((int)frame_len, (int)n_lags, false)) {
        Prof p(ctx, "autocorr_fewlags_f32");
        launch_autocorr_fewlags_f32(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, window, (int)n_lags, 0, out, nullptr);
    } else {
        Prof p(ctx, "autocorr_tiles_f32");
        launch_autocorr_tiles_f32(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, window, (int)n_lags, out);
    }
    return check_launch(ctx, __func__);
}

int vbx_normalize_f32(vbx_ctx *ctx, float *data, size_t n_rows, size_t n) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_rows == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, data && n >= 1 && n <= 0x7fffffff && n_rows <= 0x7fffffff, "bad argument");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "normalize_rows_f32"); launch_normalize_rows_f32(ctx->stream, data, (long)n_rows, (int)n); }
    return check_launch(ctx, __func__);
}

int vbx_lpc_mut_f32_wide(vbx_ctx *ctx, const float *r, size_t n_frames, size_t r_stride, size_t n_coeffs, float *out_ac,
                    float *out_kc) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_frames == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, r && out_ac, "null argument");
    VBX_REQUIRE(ctx, n_coeffs >= 1 && n_coeffs <= VBX_MAX_LPC_ORDER && r_stride >= n_coeffs + 1, "bad order / stride");
    VBX_REQUIRE(ctx, n_frames <= 0x7fffffffull, "too many rows");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "levinson_rows_f32"); launch_levinson_rows_f32(ctx->stream, r, (long)n_frames, (long)r_stride, (int)n_coeffs, out_ac, (long)n_coeffs + 1, out_kc); }
    return check_launch(ctx, __func__);
}

int vbx_autocorr_lpc_f32_wide(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                         size_t stride, const float *window, size_t n_coeffs, int normalize,
                         float *out_r, float *out_lpc) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out_r || out_lpc, "both outputs null");
    VBX_REQUIRE(ctx, n_coeffs >= 1 && n_coeffs <= VBX_MAX_LPC_ORDER && n_coeffs + 1 <= frame_len, "bad order");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    const int n_lags = (int)n_coeffs + 1;
    hipStream_t st = ctx->stream;
    if (fewlags_supported((int)frame_len, n_lags, out_lpc != nullptr)) {
        Prof p(ctx, "autocorr_lpc_f32", st);
        launch_autocorr_fewlags_f32(st, x, (long)n_frames, (int)frame_len, (long)stride, window, n_lags, normalize, out_r, out_lpc, (long)n_lags);
        return check_launch(ctx, __func__);
    }
    float *r = out_r;
    if (!r) {
        void *w = nullptr;
        rc = ws_get(ctx, vbx_ctx::WS_F32_OUT, n_frames * (size_t)n_lags * sizeof(float), &w);
        if (rc != VBX_SUCCESS) return rc;
        r = (float *)w;
    }
    if (fewlags_supported((int)frame_len, n_lags, false)) {
        Prof p(ctx, "autocorr_fewlags_f32", st);
        launch_autocorr_fewlags_f32(st, x, (long)n_frames, (int)frame_len, (long)stride, window, n_lags, 0, r, nullptr);
    } else {
        Prof p(ctx, "autocorr_tiles_f32", st);
        launch_autocorr_tiles_f32(st, x, (long)n_frames, (int)frame_len, (long)stride, window, n_lags, r);
    }
    if (normalize) { Prof p(ctx, "normalize_rows_f32", st); launch_normalize_rows_f32(st, r, (long)n_frames, n_lags); }
    if (out_lpc) { Prof p(ctx, "levinson_rows_f32", st); launch_levinson_rows_f32(st, r, (long)n_frames, n_lags, (int)n_coeffs, out_lpc, (long)n_lags); }
    return check_launch(ctx, __func__);
}

int vbx_lpc_burg_f32_wide(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                     size_t stride, const float *window, size_t n_coeffs, float *out, int32_t *status) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr, "null output");
    VBX_REQUIRE(ctx, burg_supported((int)frame_len, (int)n_coeffs), "frame_len must be in [2, 4096], order in [1, 62]");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "burg_f32"); launch_burg_f32(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, window, (int)n_coeffs, out, status); }
    return check_launch(ctx, __func__);
}

// MFCC is bound by its transforms, not by HBM: the f32 frames are widened into a dense f64 batch (windowed product
// rounded to f32 first) and take the f64 kernels; the coefficients are rounded to f32 on the way out.
int vbx_mfcc_f32(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len, size_t stride,
                 const float *window, size_t num_coeffs, double lo_hz, double hi_hz,
                 double sample_rate, float *out, int32_t *status) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr && num_coeffs >= 1, "bad argument");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    void *wi = nullptr, *wo = nullptr;
    rc = ws_get(ctx, vbx_ctx::WS_F32_IN, n_frames * frame_len * sizeof(double), &wi);
    if (rc != VBX_SUCCESS) return rc;
    rc = ws_get(ctx, vbx_ctx::WS_F32_OUT, n_frames * num_coeffs * sizeof(double), &wo);
    if (rc != VBX_SUCCESS) return rc;
    { Prof p(ctx, "widen_frames"); launch_widen_frames(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, window, (double *)wi); }
    rc = run_mfcc(ctx, ctx->stream, (const double *)wi, n_frames, frame_len, frame_len, nullptr, num_coeffs, lo_hz, hi_hz,
                  sample_rate, (double *)wo, num_coeffs, status);
    if (rc != VBX_SUCCESS) return rc;
    { Prof p(ctx, "narrow"); launch_narrow(ctx->stream, (const double *)wo, (long)(n_frames * num_coeffs), out); }
    return check_launch(ctx, __func__);
}

// Pitched::pitch at S = T = f32 (src/periodic.rs:396-455 is generic over the Sample): the frames are widened (windowed
// product rounded to f32 first), the candidates come from the f64 path and are rounded to f32 once.
int vbx_pitch_f32_wide(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len, size_t stride,
                  const float *window, float sample_rate, float threshold, float fmin, float fmax,
                  size_t kmax, vbx_pitch32 *out_cand, int32_t *out_count, int32_t *status) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out_cand != nullptr, "null output");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    void *wi = nullptr, *wo = nullptr;
    rc = ws_get(ctx, vbx_ctx::WS_F32_IN, n_frames * frame_len * sizeof(double), &wi);
    if (rc != VBX_SUCCESS) return rc;
    rc = ws_get(ctx, vbx_ctx::WS_F32_OUT, n_frames * (kmax ? kmax : 1) * sizeof(vbx_pitch), &wo);
    if (rc != VBX_SUCCESS) return rc;
    { Prof p(ctx, "widen_frames"); launch_widen_frames(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, window, (double *)wi); }
    rc = run_pitch(ctx, ctx->stream, (const double *)wi, n_frames, frame_len, frame_len, nullptr, (double)sample_rate,
                   (double)threshold, (double)fmin, (double)fmax, kmax, (vbx_pitch *)wo, 2 * kmax, out_count, status);
    if (rc != VBX_SUCCESS) return rc;
    { Prof p(ctx, "narrow"); launch_narrow(ctx->stream, (const double *)wo, (long)(n_frames * kmax * 2), (float *)out_cand); }
    return check_launch(ctx, __func__);
}


// ---- Sample = f32, reference-faithful (k_f32.hip): every fold in f32 in the reference's order ------------------------

int vbx_autocorrelate_f32(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                          size_t stride, const float *window, size_t n_lags, float *out) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr, "null output");
    VBX_REQUIRE(ctx, n_lags >= 1 && n_lags <= frame_len, "n_lags must be in [1, frame_len] (the reference panics beyond)");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "autocorr_f32_exact"); launch_autocorr_f32_exact(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, window, (int)n_lags, out); }
    return check_launch(ctx, __func__);
}

int vbx_lpc_mut_f32(vbx_ctx *ctx, const float *r, size_t n_frames, size_t r_stride, size_t n_coeffs, float *out_ac,
                    float *out_kc) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_frames == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, r && out_ac, "null argument");
    VBX_REQUIRE(ctx, n_coeffs >= 1 && n_coeffs <= VBX_MAX_LPC_ORDER && r_stride >= n_coeffs + 1, "bad order / stride");
    VBX_REQUIRE(ctx, n_frames <= 0x7fffffffull, "too many rows");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "levinson_f32_exact"); launch_levinson_f32_exact(ctx->stream, r, (long)n_frames, (long)r_stride, (int)n_coeffs, out_ac, (long)n_coeffs + 1, out_kc); }
    return check_launch(ctx, __func__);
}

int vbx_autocorr_lpc_f32(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                         size_t stride, const float *window, size_t n_coeffs, int normalize,
                         float *out_r, float *out_lpc) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out_r || out_lpc, "both outputs null");
    VBX_REQUIRE(ctx, n_coeffs >= 1 && n_coeffs <= VBX_MAX_LPC_ORDER && n_coeffs + 1 <= frame_len, "bad order");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    const int n_lags = (int)n_coeffs + 1;
    hipStream_t st = ctx->stream;
    float *r = out_r;
    if (!r) {
        void *w = nullptr;
        rc = ws_get(ctx, vbx_ctx::WS_F32_OUT, n_frames * (size_t)n_lags * sizeof(float), &w);
        if (rc != VBX_SUCCESS) return rc;
        r = (float *)w;
    }
    { Prof p(ctx, "autocorr_f32_exact", st); launch_autocorr_f32_exact(st, x, (long)n_frames, (int)frame_len, (long)stride, window, n_lags, r); }
    if (normalize) { Prof p(ctx, "normalize_rows_f32", st); launch_normalize_rows_f32(st, r, (long)n_frames, n_lags); }
    if (out_lpc) { Prof p(ctx, "levinson_f32_exact", st); launch_levinson_f32_exact(st, r, (long)n_frames, n_lags, (int)n_coeffs, out_lpc, (long)n_lags, nullptr); }
    return check_launch(ctx, __func__);
}

int vbx_lpc_burg_f32(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len,
                     size_t stride, const float *window, size_t n_coeffs, float *out, int32_t *status) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr, "null output");
    VBX_REQUIRE(ctx, burg_supported((int)frame_len, (int)n_coeffs), "frame_len must be in [2, 4096], order in [1, 62]");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    // one lane per frame, b1 / b2 in a context-owned scratch: launches of at most `chunk` frames keep it under 256 MB
    long chunk = (long)((256ull << 20) / (8ull * frame_len)) & ~63L;
    if (chunk < 64) chunk = 64;
    if ((size_t)chunk > n_frames) chunk = (long)((n_frames + 63) & ~(size_t)63);
    void *w = nullptr;
    rc = ws_get(ctx, vbx_ctx::WS_F32_IN, burg_f32_exact_scratch_bytes(chunk, (int)frame_len), &w);
    if (rc != VBX_SUCCESS) return rc;
    for (long f0 = 0; f0 < (long)n_frames; f0 += chunk) {
        const long f1 = (f0 + chunk < (long)n_frames) ? f0 + chunk : (long)n_frames;
        Prof p(ctx, "burg_f32_exact");
        launch_burg_f32_exact(ctx->stream, x, f0, f1, (long)n_frames, (int)frame_len, (long)stride, window, (int)n_coeffs, out, status, (float *)w);
    }
    return check_launch(ctx, __func__);
}

int vbx_pitch_f32(vbx_ctx *ctx, const float *x, size_t n_frames, size_t frame_len, size_t stride,
                  const float *window, float sample_rate, float threshold, float fmin, float fmax,
                  size_t kmax, vbx_pitch32 *out_cand, int32_t *out_count, int32_t *status) {
    int rc = check_frames(ctx, __func__, x, n_frames, frame_len, stride);
    if (rc != VBX_SUCCESS) return rc < 0 ? rc : VBX_SUCCESS;
    VBX_REQUIRE(ctx, out_cand != nullptr, "null output");
    VBX_REQUIRE(ctx, kmax >= 1 && kmax <= VBX_MAX_PITCH_CANDIDATES, "kmax must be in [1, VBX_MAX_PITCH_CANDIDATES]");
    VBX_REQUIRE(ctx, frame_len >= 4, "frame_len must be >= 4");
    VBX_REQUIRE(ctx, pitch_f32_exact_lds_bytes((int)frame_len, (int)kmax) + 16 <= 160 * 1024, "frame does not fit the LDS");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    auto it = ctx->lag_windows32.find(frame_len);
    if (it == ctx->lag_windows32.end()) {                   // w_lag as the f64 table of the reference's recurrence, each entry rounded to f32
        std::vector<double> h(frame_len);
        if (window_table_host(VBX_WINDOW_HANNING_LAG, frame_len, h.data()) != VBX_SUCCESS) return fail(ctx, VBX_E_INVALID, "lag window");
        std::vector<float> hf(frame_len);
        for (size_t i = 0; i < frame_len; i++) hf[i] = (float)h[i];
        float *d = nullptr;
        VBX_HIP(ctx, hipMalloc((void **)&d, frame_len * sizeof(float)));
        VBX_HIP(ctx, hipMemcpy(d, hf.data(), frame_len * sizeof(float), hipMemcpyHostToDevice));
        it = ctx->lag_windows32.emplace(frame_len, d).first;
    }
    void *wo = nullptr;
    rc = ws_get(ctx, vbx_ctx::WS_F32_OUT, n_frames * kmax * sizeof(vbx_pitch), &wo);
    if (rc != VBX_SUCCESS) return rc;
    { Prof p(ctx, "pitch_f32_exact");
      launch_pitch_f32_exact(ctx->stream, x, (long)n_frames, (int)frame_len, (long)stride, window, it->second, (double)sample_rate,
                             (double)threshold, (double)fmin, (double)fmax, (int)kmax, (pitch_t *)wo, 2 * (long)kmax, out_count, status); }
    { Prof p(ctx, "narrow"); launch_narrow(ctx->stream, (const double *)wo, (long)(n_frames * kmax * 2), (float *)out_cand); }
    return check_launch(ctx, __func__);
}

// ---- bench utility ------------------------------------------------------------------------

int vbx_synth_speech_f64(vbx_ctx *ctx, double *out, size_t n_samples, uint64_t sample_offset,
                         double sample_rate, uint64_t seed) {
    VBX_REQUIRE(ctx, ctx != nullptr, "null context");
    if (n_samples == 0) return VBX_SUCCESS;
    VBX_REQUIRE(ctx, out != nullptr && sample_rate > 0.0, "bad argument");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    { Prof p(ctx, "synth"); launch_synth(ctx->stream, out, n_samples, sample_offset, sample_rate, seed); }
    return check_launch(ctx, __func__);
}

// internal (tests only; not part of the public header): how many frames of the last FFT-path pitch / analyze call were
// handed to the direct-sum kernel because a peak decision lay inside the transforms' rounding error
int vbx_internal_last_unsure_count(vbx_ctx *ctx, int32_t *h_count) {
    VBX_REQUIRE(ctx, ctx && h_count, "null argument");
    *h_count = 0;
    if (!ctx->ws[vbx_ctx::WS_UNSURE]) return VBX_SUCCESS;
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    VBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    VBX_HIP(ctx, hipMemcpy(h_count, ctx->ws[vbx_ctx::WS_UNSURE], sizeof(int32_t), hipMemcpyDeviceToHost));
    return VBX_SUCCESS;
}

// internal (tests, bench): how many frames of the last fused analyze call the Levinson probe handed to the double-double
// recursion (k_lpc_exact.hip); -1 if the call had no LPC rows or the probe is off
int vbx_internal_last_lpc_exact_count(vbx_ctx *ctx, int32_t *h_count) {
    VBX_REQUIRE(ctx, ctx && h_count, "null argument");
    *h_count = -1;
    if (!ctx->ws[vbx_ctx::WS_LPC_LIST] || !ctx->lpc_exact) return VBX_SUCCESS;
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    VBX_HIP(ctx, hipDeviceSynchronize());
    VBX_HIP(ctx, hipMemcpy(h_count, ctx->ws[vbx_ctx::WS_LPC_LIST], sizeof(int32_t), hipMemcpyDeviceToHost));
    return VBX_SUCCESS;
}

// internal (tests only): how many frames of the last Burg / find_formants call the one-pass form's guard handed to the
// direct recursion (k_burg_fast.hip); -1 if the last such call did not take the one-pass form
int vbx_internal_last_burg_direct_count(vbx_ctx *ctx, int32_t *h_count) {
    VBX_REQUIRE(ctx, ctx && h_count, "null argument");
    *h_count = -1;
    if (!ctx->burg_list_count) return VBX_SUCCESS;
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    VBX_HIP(ctx, hipDeviceSynchronize());
    VBX_HIP(ctx, hipMemcpy(h_count, ctx->burg_list_count + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
    return VBX_SUCCESS;
}

// internal (tests only): the same count for the resonance kernel of the last find_formants call (k_roots_fast.hip)
int vbx_internal_last_roots_direct_count(vbx_ctx *ctx, int32_t *h_count) {
    VBX_REQUIRE(ctx, ctx && h_count, "null argument");
    *h_count = -1;
    if (!ctx->roots_list_count) return VBX_SUCCESS;
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    VBX_HIP(ctx, hipDeviceSynchronize());
    VBX_HIP(ctx, hipMemcpy(h_count, ctx->roots_list_count, sizeof(int32_t), hipMemcpyDeviceToHost));
    return VBX_SUCCESS;
}

// internal: cross-lane helper self-test (tests only; not part of the public header)
int vbx_selftest_lanes(vbx_ctx *ctx, double *h_out512) {   // h_out512: 1024 doubles
    VBX_REQUIRE(ctx, ctx && h_out512, "null argument");
    VBX_HIP(ctx, hipSetDevice(ctx->device));
    void *d = nullptr;
    int rc = ws_get(ctx, vbx_ctx::WS_MISC, 1024 * sizeof(double), &d);
    if (rc != VBX_SUCCESS) return rc;
    launch_selftest(ctx->stream, (double *)d);
    rc = check_launch(ctx, __func__);
    if (rc != VBX_SUCCESS) return rc;
    VBX_HIP(ctx, hipMemcpyAsync(h_out512, d, 1024 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    VBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VBX_SUCCESS;
}

}  // extern "C"
