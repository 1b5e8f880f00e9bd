#!/usr/bin/env python3
"""bench.py -- frames/sec of the vox_box hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one pass of the whole per-frame path over this rank's shard of a long synthetic
48 kHz recording (25 ms window = 1200 samples, 10 ms hop = 480 samples): Boersma pitch
candidates, autocorrelation + Levinson LPC(12), find_formants (Burg(12) -> Laguerre roots ->
resonances -> formant tracker), MFCC(13); at N > 1 it ends with the RCCL gather of the
fixed-size per-frame records to rank 0.  The audio is generated on the device before the
timed region (inputs resident in HBM), frames are range-split over ranks (weak scaling:
--hours is per GPU; the default 12.5 h/GPU is BASELINE config 5's 100 h over 8 GPUs).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel against the roof that binds it: pitch -> FP64 matrix/vector peak (flops executed,
                counted on the device); config2 / config4 / frontend -> HBM (algorithmic bytes)
  roofline_hbm  the pitch kernel against the HBM roof, as north_star asks (tiny by construction)
  cpu_baseline  the CPU oracle (C restatement of the reference path) timed on host cores
Other workloads (--workload config2|config3|config4) time a single BASELINE config.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

SR, N48, H48, P = 48000.0, 1200, 480, 12
SEG_FRAMES = 1000                 # tracker state resets every 10 s utterance
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6           # FP64 vector peak
FLOPS_PER_SINC_TERM = 13.0        # reference formula per term: sin, cos, 2 div, 9 mul/add (each counted once)
METRIC = "frames/sec (pitch+LPC+formants), 48 kHz 25 ms/10 ms hop, 1\u21928 MI355X"   # BASELINE.json "metric", verbatim
REC = 2 + 8 + 13 + 13             # per-frame record: pitch(f,s) + 4 formants(f,bw) + 13 MFCC + 13 LPC  (288 B)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hours", type=float, default=12.5, help="hours of 48 kHz audio PER GPU (pipeline / config3)")
    ap.add_argument("--frames", type=int, default=1_000_000, help="dense frames per GPU (config2 / config4)")
    ap.add_argument("--workload", default="pipeline", choices=["pipeline", "config2", "config3", "config4", "frontend"])
    ap.add_argument("--kmax", type=int, default=1, help="pitch candidates kept per frame (1 = PitchExtractor output; "
                    "64 = the reference's full sorted list, which disables the exact top-k pruning)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="wall budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (kind "port") on a bounded sample of the same workload, all host cores
# ------------------------------------------------------------------------------------------------
def cpu_baseline(workload, budget_s):
    import threading
    o = graft.load_oracle()
    pkg = graft.load_package()
    import importlib
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    cores = os.cpu_count() or 1
    if workload in ("pipeline", "config3"):
        audio = synth.synth_speech(10 * 48000 + N48, sample_offset=0)      # 10 s: 2 of 10 seconds unvoiced
        w = o.window("hanning", N48)
        n_avail = pkg.frame_count(audio.size, N48, H48)
        order = [(i * 37) % n_avail for i in range(n_avail)] * 200          # spread over the 10 s, cycled
        est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])

        def work(t):
            fr = audio[t * H48:t * H48 + N48]
            xw = fr * w
            o.pitch(xw, SR, 0.2, 75.0, 600.0, cap=4)
            if workload == "pipeline":
                o.lpc(o.autocorrelate(xw, P + 1), P)
                o.find_formants(fr, SR, P, est0)
                o.mfcc(xw, 13, 100.0, 8000.0, SR, use_fft=True)
        what = "frames spread over 10 s of the same synthetic audio (2 of 10 s unvoiced)"
    else:
        x = synth.synth_speech(4000 * 512, sample_offset=0).reshape(4000, 512)
        w = o.window("hanning", 512)
        order = list(range(4000)) * 5000
        est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])

        def work(t):
            if workload == "config2":
                o.lpc(o.autocorrelate(x[t] * w, P + 1), P)
            else:
                o.find_formants(x[t], SR, P, est0)
        what = "dense 512-sample frames of the same synthetic audio"
    done = [0] * cores
    stop_at = time.time() + budget_s
    it = iter(order)
    lock = threading.Lock()

    def runner(k):
        while time.time() < stop_at:
            with lock:
                t = next(it, None)
            if t is None:
                return
            work(t)
            done[k] += 1
    t0 = time.time()
    th = [threading.Thread(target=runner, args=(k,)) for k in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    n = sum(done)
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} {what}, {dt:.1f} s wall, {cores} threads (C restatement of the reference CPU path; "
                      "the Rust crate cannot be built here)"}


def flop_model(workload):
    """Algorithmic FP64 flops per frame of the pitch path, counted by instrumenting the oracle on a
    10 s sample: 2 * autocorrelation MACs + 13 * sinc terms."""
    o = graft.load_oracle()
    import importlib
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    audio = synth.synth_speech(10 * 48000 + N48, sample_offset=0)
    w = o.window("hanning", N48)
    idx = list(range(0, 1000, 50))
    o.counters_reset()
    for t in idx:
        o.pitch(audio[t * H48:t * H48 + N48] * w, SR, 0.2, 75.0, 600.0, cap=4)
    c = o.counters()
    n = len(idx)
    return {"autocorr_macs": c["autocorr_macs"] / n, "sinc_terms": c["sinc_terms"] / n,
            "sinc_evals": c["sinc_evals"] / n, "candidates": c["candidates"] / n,
            "flops": (2.0 * c["autocorr_macs"] + FLOPS_PER_SINC_TERM * c["sinc_terms"]) / n}


def bench_frontend(args, torch, dev, vb, vb2, pkg):
    """PCM16 ingestion -> Windower view -> RMS and pre-emphasis per frame (HBM-bound)."""
    hours = min(args.hours, 2.0)
    n = int(hours * 3600 * 48000)
    pcm = torch.randint(-32768, 32767, (n,), dtype=torch.int16, device=dev)
    audio = torch.empty(n, dtype=torch.float64, device=dev)
    F = pkg.frame_count(n, N48, H48)
    rms = torch.empty(F, dtype=torch.float64, device=dev)
    Fp = min(F, 200_000)
    pe = torch.empty((Fp, N48), dtype=torch.float64, device=dev)
    win = vb.window(pkg.WINDOW_HANNING, N48)

    def step():
        vb.pcm16_to_f64(pcm, out=audio)
        vb.L.vbx_rms_f64(vb.ctx, audio.data_ptr(), F, N48, H48, win.ptr, rms.data_ptr())
        vb.preemphasis(audio, 0.1, frame_len=N48, stride=H48, n_frames=Fp, out=pe)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    vb.profile_reset(); vb.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = vb.profile_report(); vb.profile(False)
    k = {name: ms / max(c, 1) for name, (ms, c) in prof.items()}
    ach = n * 10 / (k["pcm16"] * 1e-3) / 1e9
    out = {"metric": "frames/sec (frontend: pcm16 ingestion + rms + preemphasis)", "value": F * args.steps / dt,
           "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"frontend, {hours:g} h 48 kHz int16 PCM -> f64, rms over {F} frames, preemphasis over {Fp}"},
           "roofline": {"bound": "hbm", "kernel": "pcm16", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_sample": 10, "ms_avg": k["pcm16"]},
           "kernels_ms": {a: round(b, 3) for a, b in k.items()},
           "gbs": {"rms": F * (H48 * 8 + 8) / (k["rms"] * 1e-3) / 1e9,
                   "preemphasis": Fp * (H48 * 8 + N48 * 8) / (k["preemphasis"] * 1e-3) / 1e9}}
    print(json.dumps(out), flush=True)
    vb2.close(); vb.close()


# ------------------------------------------------------------------------------------------------
def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the product has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    pkg = graft.load_package()
    shard = pkg.shard
    # one explicit HIP stream shared by torch (allocation, cat, RCCL) and the library's kernels
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    vb = pkg.VoxBox(local, tstream.cuda_stream)
    # second context/stream: the formant chain (Burg -> roots -> latency-bound tracker scan) and the
    # small LPC / MFCC kernels run beside the FP64-bound pitch kernels instead of behind them
    sstream = torch.cuda.Stream(device=dev)
    vb2 = pkg.VoxBox(local, sstream.cuda_stream)

    wl = args.workload
    f64 = torch.float64
    if wl == "frontend":
        return bench_frontend(args, torch, dev, vb, vb2, pkg)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    if wl in ("pipeline", "config3"):
        total_frames_per_gpu = int(round(args.hours * 3600 * 100))            # 100 frames per second
        total_frames_per_gpu -= total_frames_per_gpu % SEG_FRAMES             # whole utterances per rank
        F = max(total_frames_per_gpu, SEG_FRAMES)
        lo = rank * F                                                         # weak scaling: contiguous range split
        s0, s1 = shard.sample_range(lo, lo + F, N48, H48)
        audio = torch.empty(s1 - s0, dtype=f64, device=dev)
        vb.synth_speech(s1 - s0, sample_offset=s0, sample_rate=SR, out=audio)
        frame_len, stride = N48, H48
        win = vb.window(pkg.WINDOW_HANNING, N48)
        desc = (f"{'full pitch+LPC+formants+MFCC pipeline' if wl == 'pipeline' else 'Boersma pitch path'}, "
                f"{args.hours:g} h/GPU synthetic 48 kHz, 25 ms / 10 ms hop")
    else:
        F = args.frames
        audio = torch.empty(F * 512, dtype=f64, device=dev)
        vb.synth_speech(F * 512, sample_offset=rank * F * 512, sample_rate=SR, out=audio)
        frame_len, stride = 512, 512
        win = vb.window(pkg.WINDOW_HANNING, 512)
        desc = ("batched autocorrelation + LPC order-12" if wl == "config2" else
                "LPC(Burg)->Laguerre roots->formant track") + f", {F} x 512-sample f64 frames/GPU"
    seg = np.arange(0, F, SEG_FRAMES, dtype=np.int64)

    # outputs (torch owns the device memory; the C ABI gets raw pointers)
    o_cand = torch.empty((F, args.kmax, 2), dtype=f64, device=dev)
    o_cnt = torch.empty(F, dtype=torch.int32, device=dev)
    o_pst = torch.empty(F, dtype=torch.int32, device=dev)
    o_r = torch.empty((F, P + 1), dtype=f64, device=dev)
    o_a = torch.empty((F, P + 1), dtype=f64, device=dev)
    o_form = torch.empty((F, 4, 2), dtype=f64, device=dev)
    o_fst = torch.empty(F, dtype=torch.int32, device=dev)
    o_mfcc = torch.empty((F, 13), dtype=f64, device=dev)
    o_mst = torch.empty(F, dtype=torch.int32, device=dev)
    ff = {"formants": o_form, "res": None, "count": None, "coeffs": None, "status": o_fst}
    counts = [F] * world

    arrange = os.environ.get("VBX_BENCH_ARRANGE", "two")     # stream arrangement of the pipeline workload (experiments)

    def step():
        side = vb2 if (wl == "pipeline" and arrange != "one") else vb
        if wl == "pipeline" and arrange != "one":
            sstream.wait_stream(tstream)            # the side stream starts after whatever produced the inputs
        if wl in ("pipeline", "config4"):
            side.find_formants(audio, SR, P, est0, seg_start=seg, frame_len=frame_len, stride=stride, n_frames=F, out=ff)
        if wl in ("pipeline", "config2"):
            side.autocorr_lpc(audio, P, frame_len=frame_len, stride=stride, n_frames=F, window=win, out=(o_r, o_a))
        if wl == "pipeline" and arrange not in ("mfcc_main", "mfcc_first"):
            side.mfcc(audio, 13, (100.0, 8000.0), SR, frame_len=frame_len, stride=stride, n_frames=F, window=win,
                      out=(o_mfcc, o_mst))
        if wl == "pipeline" and arrange == "mfcc_first":
            vb.mfcc(audio, 13, (100.0, 8000.0), SR, frame_len=frame_len, stride=stride, n_frames=F, window=win,
                    out=(o_mfcc, o_mst))
        if wl in ("pipeline", "config3"):
            vb.pitch(audio, SR, 0.2, 75.0, 600.0, kmax=args.kmax, frame_len=frame_len, stride=stride, n_frames=F, window=win,
                     out=(o_cand, o_cnt, o_pst))
        if wl == "pipeline" and arrange == "mfcc_main":
            vb.mfcc(audio, 13, (100.0, 8000.0), SR, frame_len=frame_len, stride=stride, n_frames=F, window=win,
                    out=(o_mfcc, o_mst))
        if wl == "pipeline" and arrange != "one":
            tstream.wait_stream(sstream)            # join before anything consumes the records
        if world > 1:   # per-frame records to rank 0 over RCCL/xGMI (no other collective on the path)
            rec = torch.cat([o_cand[:, 0, :], o_form.view(F, 8), o_mfcc, o_a], dim=1)
            shard.gather_records(rec, counts, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    for c in (vb, vb2):
        c.profile_reset()
        c.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = dict(vb.profile_report())
    prof.update(vb2.profile_report())
    vb.profile(False)
    vb2.profile(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=f64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total = F * world * args.steps
        kernels = {k: {"ms_avg": ms / max(c, 1), "launches": c} for k, (ms, c) in prof.items()}
        dom = {"pipeline": "pitch", "config3": "pitch", "config2": "autocorr_lpc", "config4": "burg"}[wl]
        dom_ms = kernels[dom]["ms_avg"]
        bytes_per_frame = {"pitch": 480 * 8 + 16, "autocorr_lpc": 512 * 8 + 2 * 13 * 8, "burg": 512 * 8 + 12 * 8 + 4}[dom]
        if dom == "autocorr_lpc" and frame_len != 512:
            bytes_per_frame = stride * 8 + 2 * 13 * 8
        ach = F * bytes_per_frame / (dom_ms * 1e-3) / 1e9
        out = {
            "metric": METRIC if wl == "pipeline" else f"frames/sec ({wl})",
            "value": total / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "frames_per_gpu": F, "frame_len": frame_len, "hop": stride,
                       "lpc_order": P, "mfcc": 13, "pitch_kmax": args.kmax, "parallelism": f"frame-range split x{world}, RCCL gather to rank 0"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_frame": bytes_per_frame, "ms_avg": dom_ms},
            "kernels_ms": {k: round(v["ms_avg"], 3) for k, v in kernels.items()},
        }
        if dom == "pitch":
            # FP64 roof with the work the kernels EXECUTED: the autocorrelation computes every lag (the oracle's
            # MAC count is exact for it); the sinc terms are counted on the device by the refine kernel itself
            # (exact top-k pruning skips most of the reference's refinements), not taken from the oracle
            fm = flop_model(wl)
            frames_w, cand_w, evals_w, terms_w = vb.profile_pitch_work()
            terms_pf = terms_w / max(frames_w, 1)
            flops_pf = 2.0 * fm["autocorr_macs"] + FLOPS_PER_SINC_TERM * terms_pf
            tf = F * flops_pf / (dom_ms * 1e-3) / 1e12
            # The pitch kernel is FP64-compute bound (autocorrelation on the FP64 matrix cores, refinement on the
            # vector ALU; both peak at 78.6 TFLOP/s on gfx950 and share the ALUs): that roof is the primary
            # `roofline`; the HBM figure north_star asks for moves to `roofline_hbm`.
            out["roofline_hbm"] = out["roofline"]
            out["roofline"] = {"bound": "mfma", "kernel": "pitch", "achieved": tf, "peak": FP64_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS, "traffic": None, "ms_avg": dom_ms,
                               "flops_per_frame": flops_pf, "autocorr_macs_per_frame": fm["autocorr_macs"],
                               "sinc_terms_per_frame": terms_pf, "sinc_evals_per_frame": evals_w / max(frames_w, 1),
                               "candidates_per_frame": cand_w / max(frames_w, 1),
                               "reference_sinc_terms_per_frame": fm["sinc_terms"],
                               "model": "FP64 flops EXECUTED: 2*autocorr MACs (every lag, v_mfma_f64_16x16x4) + 13*sinc terms "
                                        "(device counters); reference_sinc_terms = what the unpruned reference evaluates "
                                        "(oracle counters); peak = FP64 matrix = FP64 vector peak of MI355X"}
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    vb2.close()
    vb.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
