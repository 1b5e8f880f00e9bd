// vbx_comm.hip -- frame-range sharding and the one collective of the path: the gather of per-frame records to a
// destination rank, RCCL directly from C++.  The reference has no distribution at all; this is the
// MI355X-side design: one process per GPU, contiguous frame ranges, and a single grouped ncclSend/ncclRecv set in
// which every peer's payload crosses its own direct xGMI link to the destination (no ring, no reduction).
#include "../../include/voxbox_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstring>
#include <string>
#include <vector>

// shared with vbx_api.hip
extern "C" int vbx_internal_fail(vbx_ctx *ctx, int code, const char *msg);
extern "C" void *vbx_internal_stream(vbx_ctx *ctx);
extern "C" int vbx_internal_device(vbx_ctx *ctx);
extern "C" int vbx_internal_track_stitch(vbx_ctx *ctx, void *stream, vbx_resonance *formants, size_t n_frames, size_t formants_ld,
                                         size_t first, size_t stop, const double *d_state_in, int32_t *d_changed);
extern "C" double *vbx_internal_stitch_state(vbx_ctx *ctx);
extern "C" int vbx_internal_last_track_n_est(vbx_ctx *ctx);
extern "C" int vbx_internal_track_check(vbx_ctx *ctx, const vbx_resonance *formants, size_t n_frames, size_t formants_ld);

struct vbx_comm {
    ncclComm_t nccl = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;                    // transfers run here, beside the context's kernels
    hipEvent_t ready = nullptr;                      // "the records are written" (recorded on the context's stream)
    hipEvent_t done[VBX_COMM_SLOTS] = {nullptr};     // "the gather that used slot s has finished"
    hipEvent_t stitched = nullptr;                   // "the stitch has read the context's resonance rows"
    bool used[VBX_COMM_SLOTS] = {false};
};

namespace {

int fail(vbx_ctx *ctx, int code, const std::string &msg) { return vbx_internal_fail(ctx, code, msg.c_str()); }

#define VBXC_HIP(ctx, expr)                                                                        \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(ctx, VBX_E_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define VBXC_NCCL(ctx, expr)                                                                       \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return fail(ctx, VBX_E_RUNTIME, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)

std::atomic<int> g_live_comms{0};

}  // namespace

extern "C" {

int vbx_comm_live_count(void) { return g_live_comms.load(); }

int vbx_comm_unique_id(void *h_id) {
    if (!h_id) return fail(nullptr, VBX_E_INVALID, "vbx_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == VBX_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    VBXC_NCCL(nullptr, ncclGetUniqueId(&id));
    std::memcpy(h_id, &id, sizeof id);
    return VBX_SUCCESS;
}

int vbx_comm_create(vbx_ctx *ctx, const void *h_id, int world, int rank, vbx_comm **out) {
    if (!ctx || !h_id || !out || world < 1 || rank < 0 || rank >= world)
        return fail(ctx, VBX_E_INVALID, "vbx_comm_create: bad argument");
    *out = nullptr;
    const int dev = vbx_internal_device(ctx);
    VBXC_HIP(ctx, hipSetDevice(dev));
    vbx_comm *c = new vbx_comm();
    c->world = world; c->rank = rank; c->device = dev;
    ncclUniqueId id;
    std::memcpy(&id, h_id, sizeof id);
    ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) { delete c; return fail(ctx, VBX_E_RUNTIME, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->stitched, hipEventDisableTiming);
    for (int s = 0; s < VBX_COMM_SLOTS && e == hipSuccess; s++) e = hipEventCreateWithFlags(&c->done[s], hipEventDisableTiming);
    if (e != hipSuccess) { vbx_comm_destroy(c); return fail(ctx, VBX_E_RUNTIME, std::string("vbx_comm_create: ") + hipGetErrorString(e)); }
    *out = c;
    g_live_comms++;
    return VBX_SUCCESS;
}

void vbx_comm_destroy(vbx_comm *c) {
    if (!c) return;
    if (c->nccl && c->done[VBX_COMM_SLOTS - 1]) g_live_comms--;      // fully constructed communicators only
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->nccl) ncclCommDestroy(c->nccl);
    if (c->ready) hipEventDestroy(c->ready);
    if (c->stitched) hipEventDestroy(c->stitched);
    for (int s = 0; s < VBX_COMM_SLOTS; s++) if (c->done[s]) hipEventDestroy(c->done[s]);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int vbx_gather_records_f64(vbx_ctx *ctx, vbx_comm *c, const double *local, const int64_t *h_rows,
                           size_t row_doubles, int dst, double *out, int slot) {
    if (!ctx || !c || !h_rows) return fail(ctx, VBX_E_INVALID, "vbx_gather_records_f64: null argument");
    if (dst < 0 || dst >= c->world || slot < 0 || slot >= VBX_COMM_SLOTS || row_doubles < 1)
        return fail(ctx, VBX_E_INVALID, "vbx_gather_records_f64: bad dst / slot / row size");
    for (int r = 0; r < c->world; r++) if (h_rows[r] < 0) return fail(ctx, VBX_E_INVALID, "vbx_gather_records_f64: negative row count");
    const size_t mine = (size_t)h_rows[c->rank];
    if (mine > 0 && !local) return fail(ctx, VBX_E_INVALID, "vbx_gather_records_f64: null local records");
    if (c->rank == dst && !out) return fail(ctx, VBX_E_INVALID, "vbx_gather_records_f64: null output on the destination rank");
    VBXC_HIP(ctx, hipSetDevice(c->device));
    hipStream_t main = (hipStream_t)vbx_internal_stream(ctx);
    // the transfer starts when everything queued on the context's stream so far (the kernels writing `local`) is done
    VBXC_HIP(ctx, hipEventRecord(c->ready, main));
    VBXC_HIP(ctx, hipStreamWaitEvent(c->stream, c->ready, 0));
    // the transfer list (vbx_gather_plan: the same function the CPU tests drive for world 2 / 3 / 8 with uneven rows)
    std::vector<int64_t> off(c->world), cnt(c->world);
    std::vector<int32_t> op(c->world);
    if (vbx_gather_plan(h_rows, c->world, c->rank, dst, row_doubles, off.data(), cnt.data(), op.data()) != VBX_SUCCESS)
        return fail(ctx, VBX_E_INVALID, "vbx_gather_records_f64: bad row counts");
    if (op[c->rank] == VBX_GATHER_COPY && out + off[c->rank] != local)
        VBXC_HIP(ctx, hipMemcpyAsync(out + off[c->rank], local, (size_t)cnt[c->rank] * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (c->world > 1) {
        // one group: every peer -> dst transfer is posted together and runs concurrently, each over the direct
        // xGMI link between that peer and dst
        VBXC_NCCL(ctx, ncclGroupStart());
        ncclResult_t res = ncclSuccess;
        for (int r = 0; r < c->world && res == ncclSuccess; r++) {
            if (op[r] == VBX_GATHER_RECV) res = ncclRecv(out + off[r], (size_t)cnt[r], ncclDouble, r, c->nccl, c->stream);
            else if (op[r] == VBX_GATHER_SEND) res = ncclSend(local, (size_t)cnt[c->rank], ncclDouble, dst, c->nccl, c->stream);
        }
        ncclResult_t end = ncclGroupEnd();
        if (res != ncclSuccess) return fail(ctx, VBX_E_RUNTIME, std::string("ncclSend/ncclRecv: ") + ncclGetErrorString(res));
        if (end != ncclSuccess) return fail(ctx, VBX_E_RUNTIME, std::string("ncclGroupEnd: ") + ncclGetErrorString(end));
    }
    VBXC_HIP(ctx, hipEventRecord(c->done[slot], c->stream));
    c->used[slot] = true;
    return VBX_SUCCESS;
}

int vbx_comm_stitch_tracks_f64(vbx_ctx *ctx, vbx_comm *c, vbx_resonance *formants, size_t n_frames, size_t formants_ld,
                               const vbx_shard_plan_t *h_plan, int32_t *d_changed, int slot) {
    if (!ctx || !c || !h_plan) return fail(ctx, VBX_E_INVALID, "vbx_comm_stitch_tracks_f64: null argument");
    if (slot < 0 || slot >= VBX_COMM_SLOTS) return fail(ctx, VBX_E_INVALID, "vbx_comm_stitch_tracks_f64: bad slot");
    const bool recv = h_plan->continues_prev != 0, send = h_plan->continues_next != 0;
    if ((recv && c->rank == 0) || (send && c->rank == c->world - 1))
        return fail(ctx, VBX_E_INVALID, "vbx_comm_stitch_tracks_f64: the plan continues past the first / last rank");
    if ((recv || send) && (!formants || n_frames != h_plan->hi - h_plan->lo + h_plan->warm || h_plan->stop > n_frames))
        return fail(ctx, VBX_E_INVALID, "vbx_comm_stitch_tracks_f64: the rows are not the plan's frames [lo - warm, hi)");
    VBXC_HIP(ctx, hipSetDevice(c->device));
    // every host-side check BEFORE the first event or NCCL call (round-4 advisor finding): a return between this rank's
    // ncclRecv and its ncclSend would leave the next rank blocked in its matching ncclRecv.  Send-only ranks too: the row
    // they send must be the last call's.
    double *state = nullptr;
    if (recv || send) {
        int rc = vbx_internal_track_check(ctx, formants, n_frames, formants_ld);
        if (rc != VBX_SUCCESS) return rc;
        if (h_plan->warm > h_plan->stop) return fail(ctx, VBX_E_INVALID, "vbx_comm_stitch_tracks_f64: need warm <= stop");
        if (recv && !(state = vbx_internal_stitch_state(ctx)))
            return fail(ctx, VBX_E_RUNTIME, "vbx_comm_stitch_tracks_f64: no tracker state buffer on this context");
    }
    hipStream_t main = (hipStream_t)vbx_internal_stream(ctx);
    VBXC_HIP(ctx, hipEventRecord(c->ready, main));                       // the shard's own scan is done
    VBXC_HIP(ctx, hipStreamWaitEvent(c->stream, c->ready, 0));
    const int n_est = vbx_internal_last_track_n_est(ctx);
    // From the first NCCL call on, nothing returns before this rank's ncclSend is posted (round-5 advisor finding): the next rank
    // sits in the matching ncclRecv.  The first error is remembered, the send still goes out (the row it carries may then be
    // unrepaired: the caller gets the error and must abort the communicator on every rank), and the error is returned last.
    int first_rc = VBX_SUCCESS;
    auto keep = [&](int rc) { if (first_rc == VBX_SUCCESS && rc != VBX_SUCCESS) first_rc = rc; };
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess) keep(fail(ctx, VBX_E_RUNTIME, std::string("vbx_comm_stitch_tracks_f64: ") + what + ": " + hipGetErrorString(e)));
    };
    if (recv) {
        // the row the previous rank ends with: 2 n_est doubles over the direct link from rank - 1 (it sends after ITS stitch)
        VBXC_NCCL(ctx, ncclRecv(state, (size_t)(2 * n_est), ncclDouble, c->rank - 1, c->nccl, c->stream));
        keep(vbx_internal_track_stitch(ctx, (void *)c->stream, formants, n_frames, formants_ld, h_plan->warm, h_plan->stop, state, d_changed));
        // the repair reads the resonance rows in the CONTEXT's scratch, which the context's next find_formants / analyze call
        // overwrites: that call must not start before the stitch is through (a wait on the device; it costs the chain's
        // latency -- one tiny message and kernel per rank ahead -- once per step)
        hip_ok(hipEventRecord(c->stitched, c->stream), "hipEventRecord");
        hip_ok(hipStreamWaitEvent(main, c->stitched, 0), "hipStreamWaitEvent");
    } else if (d_changed) {
        hip_ok(hipMemsetAsync(d_changed, 0, sizeof(int32_t), c->stream), "hipMemsetAsync");
    }
    if (send) {
        const double *last = (const double *)formants + (n_frames - 1) * formants_ld;
        const ncclResult_t r = ncclSend(last, (size_t)(2 * n_est), ncclDouble, c->rank + 1, c->nccl, c->stream);
        if (r != ncclSuccess) keep(fail(ctx, VBX_E_RUNTIME, std::string("ncclSend: ") + ncclGetErrorString(r)));
    }
    hip_ok(hipEventRecord(c->done[slot], c->stream), "hipEventRecord");
    c->used[slot] = true;
    return first_rc;
}

int vbx_comm_wait(vbx_ctx *ctx, vbx_comm *c, int slot) {
    if (!ctx || !c || slot < 0 || slot >= VBX_COMM_SLOTS) return fail(ctx, VBX_E_INVALID, "vbx_comm_wait: bad argument");
    if (!c->used[slot]) return VBX_SUCCESS;
    VBXC_HIP(ctx, hipStreamWaitEvent((hipStream_t)vbx_internal_stream(ctx), c->done[slot], 0));
    return VBX_SUCCESS;
}

int vbx_comm_sync(vbx_comm *c) {
    if (!c) return fail(nullptr, VBX_E_INVALID, "vbx_comm_sync: null communicator");
    VBXC_HIP(nullptr, hipSetDevice(c->device));
    VBXC_HIP(nullptr, hipStreamSynchronize(c->stream));
    return VBX_SUCCESS;
}

int vbx_comm_selftest(vbx_ctx *ctx, vbx_comm *c, size_t n) {
    if (!ctx || !c || n < 1) return fail(ctx, VBX_E_INVALID, "vbx_comm_selftest: bad argument");
    VBXC_HIP(ctx, hipSetDevice(c->device));
    std::vector<double> h(n), back(n, 0.0);
    for (size_t i = 0; i < n; i++) h[i] = (double)(c->rank + 1) * 1.0e6 + (double)i * 0.25;
    double *a = nullptr, *b = nullptr;
    VBXC_HIP(ctx, hipMalloc((void **)&a, n * sizeof(double)));
    VBXC_HIP(ctx, hipMalloc((void **)&b, n * sizeof(double)));
    VBXC_HIP(ctx, hipMemcpy(a, h.data(), n * sizeof(double), hipMemcpyHostToDevice));
    VBXC_HIP(ctx, hipMemset(b, 0, n * sizeof(double)));
    ncclResult_t r1 = ncclGroupStart();
    ncclResult_t r2 = ncclSend(a, n, ncclDouble, c->rank, c->nccl, c->stream);
    ncclResult_t r3 = ncclRecv(b, n, ncclDouble, c->rank, c->nccl, c->stream);
    ncclResult_t r4 = ncclGroupEnd();
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(back.data(), b, n * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(a); hipFree(b);
    for (ncclResult_t r : {r1, r2, r3, r4})
        if (r != ncclSuccess) return fail(ctx, VBX_E_RUNTIME, std::string("vbx_comm_selftest: ") + ncclGetErrorString(r));
    if (e != hipSuccess) return fail(ctx, VBX_E_RUNTIME, std::string("vbx_comm_selftest: ") + hipGetErrorString(e));
    if (std::memcmp(h.data(), back.data(), n * sizeof(double)) != 0)
        return fail(ctx, VBX_E_RUNTIME, "vbx_comm_selftest: loopback data mismatch");
    return VBX_SUCCESS;
}

}  // extern "C"
