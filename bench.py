#!/usr/bin/env python3
"""bench.py -- frames/sec of the vox_box hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1: when no RANK is in the environment this process only LAUNCHES N fresh rank processes (before it makes any
GPU call itself), relays rank 0's JSON line and exits with their status; under torch.distributed.run (RANK set)
it is one of the ranks.  One process per GPU.  The CONTROL plane (rendezvous, the broadcast of the RCCL id, the
barrier, the max-over-ranks of the timing) runs over torch.distributed's gloo backend on CPU tensors; the DATA plane
-- the path's only exchange, the gather of per-frame records to rank 0 -- is the library's own grouped
ncclSend/ncclRecv (vbx_gather_records_f64): exactly ONE RCCL communicator per process, and no other transport exists
in the timed region (if the library's communicator does not come up on every rank the bench fails, it never falls back).

A "step" is one pass of the whole per-frame path over this rank's shard of a long synthetic 48 kHz recording
(25 ms window = 1200 samples, 10 ms hop = 480 samples): vbx_analyze_frames_f64 = Boersma pitch candidates,
autocorrelation + Levinson LPC(12) and MFCC(13) from one FFT of each frame (analyze_kernel), find_formants (Burg(12) ->
Laguerre roots -> resonances -> formant tracker) beside it, written as one fixed-size record per frame; at N > 1 the
step ends by queueing the gather of the records to rank 0 (it overlaps the next step's kernels; every gather is complete
when the timed region ends).  The audio is generated on the device before the timed region (inputs resident in HBM);
frames are range-split over ranks (weak scaling: --hours is per GPU; the default 12.5 h/GPU is BASELINE config 5's 100 h
over 8 GPUs).

Rank 0 prints ONE JSON line under 6 KB (compact_line: the driver keeps the last 8 KB of stdout) with the contract fields plus
  roofline        the kernel with the largest measured time against the roof that binds it.  pitch / analyze -> FP64
                  vector peak with the flops the kernel EXECUTES (`frac`), `issue_frac` = the share of SIMD time the vector
                  ALU is issuing; everything else -> HBM, algorithmic bytes.  `traffic` = measured HBM bytes per launch;
                  for the headline kernel `traffic`, `issue_frac` and `valu_insts_per_frame` are MEASURED IN THE RUN (three
                  short `rocprofv3 --pmc` child passes, live_traffic()), elsewhere they come from the committed PMC passes
                  (profiles/pmc_traffic.json): `traffic_source` says which
  roofline_hbm    the pitch kernel against the HBM roof, as north_star asks (tiny by construction)
  cpu_baseline    the CPU oracle (C restatement of the reference path) timed natively (oracle/vbx_cpu_bench.c) on
                  1 core and on all the cores the process may use, at the GPU leg's frame geometry
  sub_benchmarks  (default run, N = 1) {name: frames/s}: BASELINE configs 2, 3 (kmax 1 / 8 / the whole candidate Vec), 4, fourteen
                  other frame shapes, the pipeline on real 44.1 kHz speech, config 5 whole on one GPU -- a few steps each
                  outside the headline's timed region
and writes the FULL record (every sub-benchmark's kernels and rooflines, per-shape tables, the parity note, the model strings) to
--detail (default gpurun_out/bench_detail.json).
Other workloads (--workload config2|config3|config4|frontend) time a single BASELINE config; --frame-len / --hop move the
pipeline and config3 to another frame shape (2048 / 1024 is the reference example's), under its own metric name.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

SR, N48, H48, P = 48000.0, 1200, 480, 12
SEG_FRAMES = 1000                 # tracker state resets every 10 s utterance
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6           # FP64 vector peak (= FP64 matrix peak on gfx950)
FLOPS_PER_SINC_TERM = 13.0        # reference formula per term: sin, cos, 2 div, 9 mul/add (each counted once)
METRIC = "frames/sec (pitch+LPC+formants), 48 kHz 25 ms/10 ms hop, 1→8 MI355X"   # BASELINE.json "metric", verbatim
CONFIG3_FRAMES = 3_599_998        # 10 h at 48 kHz / 1200 / 480

# algorithmic HBM bytes per frame of each kernel, keyed by the profile name
ALG_BYTES = {
    "pitch": lambda n, hop, p: hop * 8 + 16,                      # new samples in, top candidate out
    "analyze": lambda n, hop, p: hop * 8 + 16 + 13 * 8 + 13 * 8,  # fused: + MFCC + LPC rows out
    "autocorr_lpc": lambda n, hop, p: hop * 8 + 2 * (p + 1) * 8,
    "burg": lambda n, hop, p: hop * 8 + p * 8 + 4,                # the direct recursion (VBX_BURG_DIRECT=1; orders other than 12)
    "burg_lags": lambda n, hop, p: hop * 8 + 3 * (p + 1) * 8,     # one-pass Burg: new samples in, lag sums + edge samples out
    "burg_recursion": lambda n, hop, p: 3 * (p + 1) * 8 + p * 8 + 4,
    "burg_direct_list": lambda n, hop, p: hop * 8 + p * 8 + 4,    # per frame ON THE LIST (~1 % of the batch)
    "formant_resonances": lambda n, hop, p: p * 8 + 4 + 32 * 16 + 4,
    "tracker": lambda n, hop, p: (p // 2) * 16 + 8 + 64,             # a row is read up to its count (<= p / 2 resonances), not all 32 slots
    "tracker_chunked": lambda n, hop, p: 2 * ((p // 2) * 16 + 8) + 64,   # warm-up: every row is read twice
    "mfcc": lambda n, hop, p: hop * 8 + 13 * 8,
    "pcm16": lambda n, hop, p: 10,
    "lpc_rows": lambda n, hop, p: 2 * (p + 1) * 8 + 2 * 13 * 8,    # lag sums in, coefficients out (in place in the record's LPC row) + the MFCC row's deferred tail
    "mfcc_rows": lambda n, hop, p: 2 * 13 * 8,                    # filter sums in, coefficients out (in place in the record's MFCC row)
    "lpc_exact_list": lambda n, hop, p: hop * 8 + (p + 1) * 8,    # per frame ON THE LIST (~0.1 % of the synthetic signal's frames)
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hours", type=float, default=12.5, help="hours of 48 kHz audio PER GPU (pipeline / config3)")
    ap.add_argument("--frames", type=int, default=1_000_000, help="dense frames per GPU (config2 / config4)")
    ap.add_argument("--frame-len", type=int, default=N48, help="pipeline / config3: samples per frame (default: BASELINE's 25 ms "
                    "at 48 kHz; 2048 with --hop 1024 is the shape of the reference's examples/pitch_detection.rs)")
    ap.add_argument("--hop", type=int, default=H48, help="pipeline / config3: samples between frames")
    ap.add_argument("--workload", default="pipeline", choices=["pipeline", "config2", "config3", "config4", "frontend"])
    ap.add_argument("--kmax", type=int, default=1, help="pitch candidates kept per frame (1 = PitchExtractor output; "
                    "64 = the head of the reference's sorted list, which disables the exact top-k pruning)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="wall budget of the CPU baseline leg (both legs together)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-sub", action="store_true", help="skip the sub_benchmarks of the default run (configs 2, 3, 4)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure the headline kernel's HBM traffic in this run (two short "
                    "rocprofv3 --pmc child passes, FETCH_SIZE and WRITE_SIZE, ~10 s each): roofline.traffic then comes from the committed "
                    "profiles/pmc_traffic.json")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"), help="where the FULL record goes (per-shape "
                    "tables, per-kernel times of every sub-benchmark, the parity note, the model strings); stdout carries only the "
                    "compact line (< 6 KB: the driver keeps the last 8 KB of stdout)")
    ap.add_argument("--signal", default="synthetic", choices=["synthetic", "speech"], help="speech: the pipeline on a recording of REAL "
                    "speech (tests/golden/sample-two_vowels.wav, 44.1 kHz, tiled with per-tile gains and a -70 dB dither) at the "
                    "shapes a 44.1 kHz caller uses (1103 / 441 = 25 ms / 10 ms; 1024 / 512 = tests/lib.rs:56-57), order 13 "
                    "(examples/formant_extraction/src/main.rs:53), beside the synthetic signal at the same shapes; its own metric "
                    "name, never the headline")
    ap.add_argument("--host-fed", action="store_true", help="pipeline only: the recording starts in pinned HOST memory as 16-bit PCM "
                    "and crosses PCIe inside the timed region, in chunks, double-buffered against the kernels (vbx_analyze_frames_pcm16); "
                    "reported under its own metric name, never as the headline `value`")
    ap.add_argument("--chunk-frames", type=int, default=250_000, help="--host-fed: frames per H2D chunk (whole utterances)")
    ap.add_argument("--utterance-frames", type=int, default=0, help="pipeline / config3: frames per utterance (the formant tracker "
                    "restarts from the initial estimates at every utterance start).  0 (default): the WHOLE recording is one "
                    "utterance, what the reference's user loop over a file is (tests/lib.rs:75-79) -- at N > 1 the even frame split "
                    "cuts it and the track is carried across the rank boundaries (vbx_comm_stitch_tracks_f64)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves (this process never touches the GPU)
# ------------------------------------------------------------------------------------------------
def launch_ranks(args):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    return supervise(procs)


def supervise(procs, grace_s=10.0, poll_s=0.2, relay=sys.stdout):
    """Wait for the rank processes.  If one dies (or the watchdog ends it: exit code 5) the others would sit in a rendezvous or
    a collective until its timeout, so its exit ends them too: terminate(), and after `grace_s` seconds kill() whatever ignored
    that (a rank stuck inside a RCCL kernel does not run Python signal handlers).  These are exactly the children started by
    launch_ranks.  Relays rank 0's stdout; returns 0 only if every rank exited 0."""
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.monotonic() + grace_s
            while any(p.poll() is None for p in procs) and time.monotonic() < deadline:
                time.sleep(poll_s)
            for r, p in enumerate(procs):
                if p.poll() is None:
                    sys.stderr.write(f"bench.py: rank {r} ignored terminate() for {grace_s:.0f} s: kill()\n")
                    p.kill()
            break
        time.sleep(poll_s)
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    codes = [p.wait() for p in procs]
    if out0 and relay is not None:
        relay.write(out0)
        relay.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or failed:
        sys.stderr.write(f"bench.py: ranks failed (rank, exit code): {bad}\n")
        return 1
    return 0


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (kind "port") on a bounded sample of the same workload, natively threaded
# ------------------------------------------------------------------------------------------------
def cpu_baseline(workload, budget_s, frame_len=N48, hop=H48):
    """`hop` is the GPU leg's stride: 480 for the Windower view of the pipeline / config 3, 512 (dense frames) for
    configs 2 and 4."""
    import importlib
    o = graft.load_oracle()
    graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    logical = os.cpu_count() or 1
    usable = o.usable_cores()          # affinity mask capped by the cgroup CPU quota
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:    # physical cores: distinct (package, core) pairs
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        physical = len(pairs) or None
    except OSError:
        physical = None
    if workload in ("pipeline", "config3"):
        audio = synth.synth_speech(10 * 48000 + frame_len, sample_offset=0)  # 10 s: 2 of 10 seconds unvoiced
        what = (f"frames of the 48 kHz / {frame_len}-sample / {hop}-sample-hop view of 10 s of the same synthetic audio "
                "(2 of 10 s unvoiced), scrambled order")
    else:
        audio = synth.synth_speech(40 * 48000 + frame_len, sample_offset=0)
        what = f"{frame_len}-sample frames (hop {hop}, the GPU leg's stride) of 40 s of the same synthetic audio, scrambled order"
    leg = max(budget_s / 2.0, 1.0)
    n1, t1 = o.cpu_bench(workload, audio, frame_len, hop, P, SR, 1, leg)
    na, ta = o.cpu_bench(workload, audio, frame_len, hop, P, SR, usable, leg)
    return {"value": na / ta, "unit": "frames/s", "cores": usable, "kind": "port",
            "one_core": {"value": n1 / t1, "unit": "frames/s", "cores": 1, "frames": n1, "seconds": round(t1, 2)},
            "logical_cpus": logical, "physical_cores": physical, "cgroup_cpu_quota": quota,
            "sample": f"{na} {what}, {ta:.1f} s wall on {usable} native threads (and {n1} frames in {t1:.1f} s on 1 thread); "
                      "oracle/vbx_cpu_bench.c: C restatement of the reference CPU path, pthreads over frames -- the Rust "
                      "crate itself is single-threaded and cannot be built here"}


_FLOP_MODEL = {}


def flop_model(frame_len=N48, hop=H48):
    """FP64 flops per frame of the reference's pitch path, counted by instrumenting the oracle on a
    10 s sample: 2 * autocorrelation MACs + 13 * sinc terms."""
    if (frame_len, hop) in _FLOP_MODEL:
        return _FLOP_MODEL[(frame_len, hop)]
    import importlib
    o = graft.load_oracle()
    graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    audio = synth.synth_speech(10 * 48000 + frame_len, sample_offset=0)
    w = o.window("hanning", frame_len)
    n_fr = (10 * 48000) // hop
    idx = list(range(0, n_fr, max(1, n_fr // 20)))
    o.counters_reset()
    for t in idx:
        o.pitch(audio[t * hop:t * hop + frame_len] * w, SR, 0.2, 75.0, 600.0, cap=4)
    c = o.counters()
    n = len(idx)
    m = {"autocorr_macs": c["autocorr_macs"] / n, "sinc_terms": c["sinc_terms"] / n,
         "sinc_evals": c["sinc_evals"] / n, "candidates": c["candidates"] / n,
         "flops": (2.0 * c["autocorr_macs"] + FLOPS_PER_SINC_TERM * c["sinc_terms"]) / n}
    _FLOP_MODEL[(frame_len, hop)] = m
    return m


_PMC = None


def pmc_entry(kernel):
    """One kernel's entry of profiles/pmc_traffic.json (committed PMC passes: HBM bytes per frame = FETCH_SIZE x2 per the
    guide's gfx950 correction + WRITE_SIZE; SQ counters per frame where a pass collected them), or None."""
    global _PMC
    if _PMC is None:
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                _PMC = json.load(f)
        except (OSError, ValueError):
            _PMC = {}
    return _PMC.get(kernel)


def measured_traffic(kernel, frames):
    e = pmc_entry(kernel)
    if not e:
        return None, None
    return e["bytes_per_frame"] * frames, {k: e[k] for k in ("bytes_per_frame", "source", "commit") if k in e}


# kernel-name prefixes of the dominant kernels whose traffic the default run measures itself (rocprofv3's counter CSV: Kernel_Name)
LIVE_KERNELS = {"analyze": "void vbx::analyze_kernel<true, true, true, 0,"}


def counter_values(out_dir, prefix, names):
    """{counter: value} of the ONE launch of the kernel whose name starts with `prefix` in rocprofv3's *counter_collection.csv files under
    out_dir (columns Kernel_Name, Counter_Name, Counter_Value: one row per dispatch and counter).  Anything but exactly one launch is an
    error: a kernel renamed in the sources must not silently read as zero traffic."""
    import csv
    import glob
    out = {}
    for name in names:
        vals = []
        for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if r["Kernel_Name"].startswith(prefix) and r["Counter_Name"] == name:
                        vals.append(float(r["Counter_Value"]))
        if len(vals) != 1:
            raise RuntimeError(f"{name}: expected one launch of '{prefix}', found {len(vals)}")
        out[name] = vals[0]
    return out


def live_traffic(dom, hours=0.5, timeout_s=45.0):
    """HBM bytes per frame (and the vector ALU's busy share) of the dominant kernel, MEASURED IN THIS RUN (round 6; the round-5 review: a figure read from a committed
    file can never be refuted by a driver line): three child processes, each `rocprofv3 --pmc <one counter; the third: one SQ group> -- python3 bench.py --hours
    0.5 --steps 1 --warmup 0 --no-cpu --no-sub --no-live-traffic` (separate passes, counters only -- no trace domains --, as
    MI355X_MICROARCH.md prescribes; the children are fresh processes, started from /tmp), FETCH_SIZE x 1024 x 2 (gfx950 tallies a 128-B
    request of a streaming read at 64 B) + WRITE_SIZE x 1024 of the kernel's ONE launch over the child's 180,000 frames.
    Returns (bytes_per_frame, source dict) or raises."""
    import shutil
    import tempfile
    prefix = LIVE_KERNELS[dom]
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    got, frames = {}, None
    tmp = tempfile.mkdtemp(prefix="vbx_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"):
            out_dir = os.path.join(tmp, counter.split()[0])
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VBX_BENCH_SELFCHECK"):
                env.pop(k, None)
            cmd = [exe, "--pmc"] + counter.split() + ["--output-format", "csv", "-d", out_dir, "--", sys.executable, os.path.abspath(__file__),
                   "--hours", str(hours), "--steps", "1", "--warmup", "0", "--no-cpu", "--no-sub", "--no-live-traffic",
                   "--detail", os.path.join(tmp, counter.split()[0] + "_detail.json")]
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            line = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
            if p.returncode != 0 or not line:
                raise RuntimeError(f"rocprofv3 --pmc {counter} child failed (rc {p.returncode}): {p.stderr[-300:]}")
            frames = json.loads(line[-1])["config"]["frames_per_gpu"]
            got.update(counter_values(out_dir, prefix, counter.split()))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch_b, write_b = got["FETCH_SIZE"] * 1024.0 * 2.0, got["WRITE_SIZE"] * 1024.0
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over every SIMD; GRBM_GUI_ACTIVE is summed over the 8 XCDs (tools/prof_commit.py)
    simd_quads = got["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0
    return (fetch_b + write_b) / frames, {
        "valu_busy": got["SQ_ACTIVE_INST_VALU"] / simd_quads, "valu_insts_per_frame": got["SQ_INSTS_VALU"] / max(got["SQ_WAVES"], 1.0),
        "bytes_per_frame": (fetch_b + write_b) / frames, "fetch_bytes_per_frame": fetch_b / frames, "write_bytes_per_frame": write_b / frames,
        "source": f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate child passes of this bench, {hours:g} h = {frames} frames, one launch of "
                  f"{prefix}...>; FETCH_SIZE x1024 x2, WRITE_SIZE x1024)", "commit": "this run"}


def traffic_key(wl, dom, frame_len, stride):
    """profiles/pmc_traffic.json key of the dominant kernel of a workload (tests/test_bench_contract.py walks every key this
    function can return for the shapes the bench documents)."""
    if wl == "config2":
        return "autocorr_lpc_512"
    if wl == "config4":
        return dom + "_512"
    if wl == "frontend":
        return "pcm16"
    return dom if (frame_len, stride) == (N48, H48) else f"{dom}_{frame_len}"


# ------------------------------------------------------------------------------------------------
# one timed workload: K steps bracketed by a fence, HIP events around every kernel launch
# ------------------------------------------------------------------------------------------------
def timed(vb, torch, step, warmup, steps, barrier=None):
    def fence():
        torch.cuda.synchronize()                                              # every stream of the device, the transfers included
        if barrier is not None:
            barrier()
    for i in range(warmup):
        step(i)
    fence()
    vb.profile_reset()
    vb.profile(True)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if barrier is not None:
        barrier()
    prof = dict(vb.profile_report())
    PROF_STREAMS.clear()
    PROF_STREAMS.update(vb.profile_streams())
    work = vb.profile_pitch_work()
    vb.profile(False)
    return dt, prof, work


PROF_STREAMS = {}      # kernel name -> stream of the last timed() call (vbx_profile_stream): 0 = the context's stream


def dominant_kernel(prof):
    """The kernel with the largest measured time ON THE CRITICAL STREAM (the context's: the step ends when it does).
    Kernels of the side / tracker streams run beside it; their HIP-event times include the time they spend co-resident
    with the critical stream's kernel (burg_lags beside analyze: 9-12 ms of events for 1.8 ms of work) and do not rank."""
    total = lambda k: prof[k][0]
    crit = [k for k in prof if PROF_STREAMS.get(k, 0) == 0]
    return max(crit or list(prof), key=total)


def roofline_for(wl, prof, work, F, frame_len, stride, steps):
    """The dominant kernel (largest measured time) against its roof.  Returns (roofline, roofline_hbm or None, kernels_ms)."""
    kernels = {k: {"ms_avg": ms / max(c, 1), "launches": c} for k, (ms, c) in prof.items()}
    dom = dominant_kernel(prof)                                          # largest measured time on the critical stream
    # a kernel may run in several launches per step (the time slices of find_formants): per-launch figures
    per_step = max(kernels[dom]["launches"] // max(steps, 1), 1)
    Fl = F / per_step                                                    # frames per launch
    dom_ms = kernels[dom]["ms_avg"]
    bytes_per_frame = ALG_BYTES.get(dom, lambda n, hop, p: hop * 8)(frame_len, stride, P)
    ach = Fl * bytes_per_frame / (dom_ms * 1e-3) / 1e9
    tkey = traffic_key(wl, dom, frame_len, stride)
    traffic, tsrc = measured_traffic(tkey, Fl)
    hbm = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
           "algorithmic_bytes_per_frame": bytes_per_frame, "ms_avg": dom_ms, "frames_per_launch": Fl,
           "launches_per_step": per_step,
           "note": "dominant kernel = largest measured time on the context's stream (the critical path; kernels of the side streams "
                   "run beside it); ms_avg from HIP events on the stream the kernel runs on"}
    kms = {k: round(v["ms_avg"], 3) for k, v in kernels.items()}
    if dom not in ("pitch", "analyze"):
        return hbm, None, kms, kernels
    # FP64 roof (vector peak = matrix peak on gfx950: 78.6 TFLOP/s).
    #   `achieved` / `frac` (the headline): the arithmetic the kernel EXECUTES -- the autocorrelation is two real FFTs of
    #     the zero-padded frame (5 N log2 N / 2 flops each, + the power spectrum), not N^2 MACs -- plus 13 flops for every
    #     sinc term it evaluated (device counters; the exact top-k pruning skips most of the reference's refinements and
    #     pruned work is not credited).  Small by construction: the kernel is bound by vector-instruction ISSUE, most of
    #     which is not FMA work (Brent scalars, selects, address arithmetic) -- `issue_frac` says how busy the issue port is.
    #   `reference_sums_at_peak`: the same kernel time against what the REFERENCE's algorithm would need at the FP64 peak
    #     (2 * the MACs of its O(N^2) all-lag autocorrelation + the same sinc terms).  Above 1 it means the kernel finishes
    #     sooner than the reference's lag sums could at peak; it is a speed-up statement, not a roofline fraction.
    fm = flop_model(frame_len, stride)
    frames_w, cand_w, evals_w, terms_w = work
    terms_pf = terms_w / max(frames_w, 1)
    nc = (1024 if 512 <= frame_len <= 1024 else 1200 if 1024 < frame_len <= 1200 else 2048 if 1200 < frame_len <= 2048 else
          4096 if 2048 < frame_len <= 4096 else 0)                      # complex FFT length of the frame's plan
    nfft = 2.0 * nc
    fft_flops = (2.0 * 2.5 * nfft * np.log2(nfft) + 6.0 * nfft) if nc else 0.0   # two real transforms (half the complex cost) + |X|^2, split
    exec_pf = (fft_flops if nc else 2.0 * fm["autocorr_macs"]) + FLOPS_PER_SINC_TERM * terms_pf
    tfe = Fl * exec_pf / (dom_ms * 1e-3) / 1e12
    ref_pf = 2.0 * fm["autocorr_macs"] + FLOPS_PER_SINC_TERM * terms_pf
    tfr = Fl * ref_pf / (dom_ms * 1e-3) / 1e12
    e = pmc_entry(tkey) or {}
    roof = {"bound": "fp64_valu" if nc else "mfma", "kernel": dom, "achieved": tfe, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tfe / FP64_PEAK_TFLOPS,
            "issue_frac": e.get("valu_busy"), "issue_frac_source": e.get("sq_source"),
            "traffic": traffic, "traffic_source": tsrc, "ms_avg": dom_ms, "frames_per_launch": Fl,
            "flops_per_frame": exec_pf, "sinc_terms_per_frame": terms_pf, "sinc_evals_per_frame": evals_w / max(frames_w, 1),
            "candidates_per_frame": cand_w / max(frames_w, 1), "reference_sinc_terms_per_frame": fm["sinc_terms"],
            "model": (f"executed FP64 flops: 2 real FFTs of {int(nfft)} points (5 N log2 N / 2 each) + power spectrum + 13 * sinc "
                      "terms evaluated (device counters); peak = FP64 vector = FP64 matrix peak (78.6 TFLOP/s; the kernel issues "
                      "no MFMA); the kernel is vector-ISSUE bound, not FMA bound: see issue_frac") if nc else
                     "executed FP64 flops: 2 * the all-lag autocorrelation MACs (matrix-core tiles) + 13 * sinc terms evaluated",
            "reference_sums_at_peak": {"flops_per_frame": ref_pf, "autocorr_macs_per_frame": fm["autocorr_macs"], "achieved": tfr,
                                       "ratio": tfr / FP64_PEAK_TFLOPS,
                                       "meaning": "kernel time against running the reference's O(N^2) lag sums (+ the same sinc terms) "
                                                  "at the FP64 peak; > 1 = faster than that; NOT a roofline fraction"}}
    return roof, hbm, kms, kernels


PARITY_NOTE = {
    "checked_against": "oracle/vbx_oracle.c (C restatement of the reference, pinned by its 26 inline known-answer tests "
                       "and both WAV fixtures: tests/test_oracle_kat.py), through the C ABI in tests/ -m gpu; "
                       "tests/test_gpu_soak.py holds 50,000 consecutive frames of this workload (and 20,000 each of configs 2 "
                       "and 4) to the oracle on every run of the suite",
    "unpinned_by_the_reference": ["MFCC values (rustfft un-vendored; the reference asserts finiteness only)",
                                  "find_formants end to end (the reference's test prints)",
                                  "sinc / Brent values beyond the one 150 Hz vector (1e-2 Hz)",
                                  "sample 0.10 window phase recurrence and the linear resampler (crate un-vendored)"],
    "designed_approximations": ["MFCC inside the fused call at frame lengths that do not divide its transform (not this workload's 1200): "
                                "the frame's DFT bins are interpolated from the transform's by 24-40 taps, design error < 1e-14 of "
                                "the largest bin (tests/test_mfcc_interp_table.py, exact DFT on the CPU); MFCC values within 1e-11 "
                                "of the chirp-z kernel's exact arithmetic (tests/test_gpu_analyze.py), 1e-6 of the oracle's"]}


def config4_whole(F, step_s, kernels, steps, n=512, p=P):
    """Config 4 as a whole against both roofs.  SURVEY 8d's figures are the REFERENCE's arithmetic: 4264 B and ~135 kflop per
    frame (Burg 10 N p + Laguerre (p - 2) 20 3p complex MACs).  The kernels no longer execute that: the one-pass Burg does
    2 N (p + 1) flops of lag sums + ~6 p^2 of recursion (+ ~1 % of the frames redone directly), the conjugate-pair root
    finder ~5.7 solves x ~4.2 iterations x (three real synthetic divisions: 12 p flops + ~40 of Laguerre step) + deflation +
    the Newton polish (~5.5 kflop at p = 12).  `fp64_frac` is on the EXECUTED flops; the reference's figure is kept beside it
    as a speed-up statement, as in the headline line."""
    executed = 2.0 * n * (p + 1) + 6.0 * p * p + 0.01 * 10.0 * n * p + (5.7 * 4.2 * (12.0 * p + 40.0) + 6 * 4.0 * p + 12 * 4.0 * p)
    return {"bytes_per_frame": 4264, "GBps": F * 4264 / step_s / 1e9, "hbm_frac": F * 4264 / step_s / 1e9 / HBM_PEAK_GBS,
            "flops_per_frame": executed, "TFLOPs": F * executed / step_s / 1e12,
            "fp64_frac": F * executed / step_s / 1e12 / FP64_PEAK_TFLOPS,
            "reference_arithmetic_at_peak": {"flops_per_frame": 135e3, "ratio": F * 135e3 / step_s / 1e12 / FP64_PEAK_TFLOPS,
                                             "meaning": "step time against running the reference's per-order Burg sweeps and 20-iteration "
                                                        "complex Laguerre at the FP64 peak; NOT a roofline fraction"},
            "kernels_ms_per_step": {k: round(v["ms_avg"] * v["launches"] / steps, 3) for k, v in kernels.items()}}


def bench_frontend(args, torch, dev, vb, pkg):
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

    def step(i):
        vb.pcm16_to_f64(pcm, out=audio)
        vb.L.vbx_rms_f64(vb.ctx, audio.data_ptr(), F, N48, H48, win.ptr, rms.data_ptr())
        vb.preemphasis(audio, 0.1, frame_len=N48, stride=H48, n_frames=Fp, out=pe)
    dt, prof, _ = timed(vb, torch, step, args.warmup, args.steps)
    k = {name: ms / max(c, 1) for name, (ms, c) in prof.items()}
    ach = n * 10 / (k["pcm16"] * 1e-3) / 1e9
    traffic, tsrc = measured_traffic("pcm16", n)
    out = {"metric": "frames/sec (frontend: pcm16 ingestion + rms + preemphasis)", "value": F * args.steps / dt,
           "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"frontend, {hours:g} h 48 kHz int16 PCM -> f64, rms over {F} frames, preemphasis over {Fp}",
                      "samples": n, "frames_per_gpu": F},
           "roofline": {"bound": "hbm", "kernel": "pcm16", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                        "algorithmic_bytes_per_sample": 10, "ms_avg": k["pcm16"]},
           "kernels_ms": {a: round(b, 3) for a, b in k.items()},
           "gbs": {"rms": F * (H48 * 8 + 8) / (k["rms"] * 1e-3) / 1e9,
                   "preemphasis": Fp * (H48 * 8 + N48 * 8) / (k["preemphasis"] * 1e-3) / 1e9}}
    emit(out, args.detail)
    vb.close()


# ------------------------------------------------------------------------------------------------
# host-fed operation: 16-bit PCM in pinned host memory -> PCIe -> the fused frame loop, chunked and double-buffered
# ------------------------------------------------------------------------------------------------
def bench_host_fed(args, torch, dev, vb, pkg):
    """What a deployment that does not keep the audio on the device pays: the PCIe-inclusive rate.  The recording sits in
    pinned host memory as 16-bit PCM (960 B of new samples per frame: 63 GB/s of link carry 66 M frames/s, above what the
    GPU analyses); chunk c+1's copy runs on a copy stream while chunk c is analysed (vbx_analyze_frames_pcm16, which
    widens in registers), two device buffers.  Records stay on the device (as in the resident bench)."""
    N, H = N48, H48
    C = max(SEG_FRAMES, args.chunk_frames - args.chunk_frames % SEG_FRAMES)
    F = int(round(args.hours * 3600 * SR / H)); F -= F % C; F = max(F, C)
    n_chunks = F // C
    ns_chunk = (C - 1) * H + N
    # build the PCM recording on the device (synthetic speech quantised to int16), park it in pinned host memory
    host = torch.empty((F - 1) * H + N, dtype=torch.int16).pin_memory()
    tmp = torch.empty(ns_chunk, dtype=torch.float64, device=dev)
    for c in range(n_chunks):
        vb.synth_speech(ns_chunk, sample_offset=c * C * H, sample_rate=SR, out=tmp)
        torch.cuda.synchronize()
        host[c * C * H:c * C * H + ns_chunk] = torch.clamp(torch.round(tmp * (0.9 * 32767.0 / 0.5)), -32768, 32767).to(torch.int16).cpu()
    del tmp
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0, mfcc=(13, 100.0, 8000.0))
    REC = int(vb.L.vbx_record_doubles(params))
    rec = torch.empty((F, REC), dtype=torch.float64, device=dev)
    st3 = torch.empty((3, C), dtype=torch.int32, device=dev)
    bufs = [torch.empty(ns_chunk, dtype=torch.int16, device=dev) for _ in range(2)]
    seg = np.arange(0, C, SEG_FRAMES, dtype=np.int64)
    main = torch.cuda.current_stream()
    copy = torch.cuda.Stream(device=dev)
    ready = [torch.cuda.Event() for _ in range(2)]          # chunk landed in bufs[b]
    freed = [torch.cuda.Event() for _ in range(2)]          # the kernels are done with bufs[b]

    def upload(c):
        b = c % 2
        with torch.cuda.stream(copy):
            copy.wait_event(freed[b])
            bufs[b].copy_(host[c * C * H:c * C * H + ns_chunk], non_blocking=True)
            ready[b].record(copy)

    def step(i):
        for b in range(2):
            freed[b].record(main)
        upload(0)
        for c in range(n_chunks):
            b = c % 2
            if c + 1 < n_chunks:
                upload(c + 1)
            main.wait_event(ready[b])
            vb.analyze_frames_pcm16(bufs[b], params, seg_start=seg, frame_len=N, stride=H, n_frames=C,
                                    out=rec[c * C:(c + 1) * C], record_ld=REC, status=st3)
            freed[b].record(main)
    dt, prof, work = timed(vb, torch, step, args.warmup, args.steps)
    # the same chunks already resident (no copy in the loop): what the link costs
    def step_resident(i):
        for c in range(n_chunks):
            vb.analyze_frames_pcm16(bufs[c % 2], params, seg_start=seg, frame_len=N, stride=H, n_frames=C,
                                    out=rec[c * C:(c + 1) * C], record_ld=REC, status=st3)
    dt_res, _, _ = timed(vb, torch, step_resident, 1, args.steps)
    # the copies alone
    t0 = time.perf_counter()
    for c in range(n_chunks):
        bufs[c % 2].copy_(host[c * C * H:c * C * H + ns_chunk], non_blocking=True)
    torch.cuda.synchronize()
    dt_copy = time.perf_counter() - t0
    kms = {k: round(ms / max(c, 1), 3) for k, (ms, c) in prof.items()}
    out = {"metric": "frames/sec (host-fed: 16-bit PCM in pinned host memory -> PCIe -> pitch+LPC+formants+MFCC), 48 kHz 25 ms/10 ms hop",
           "value": F * args.steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"host-fed pipeline, {F} frames ({F * H / SR / 3600:.2f} h) as int16 PCM in pinned host memory, "
                                  f"chunks of {C} frames, two device buffers, copy stream beside the kernels",
                      "frames_per_gpu": F, "frame_len": N, "hop": H, "chunk_frames": C, "pcm_bytes_per_frame": 2 * H},
           "pcie": {"bytes_per_step": int(2 * ns_chunk * n_chunks), "copy_alone_GBps": 2 * ns_chunk * n_chunks / dt_copy / 1e9,
                    "link_peak_GBps": 63.0, "inclusive_GBps": 2 * ns_chunk * n_chunks * args.steps / dt / 1e9},
           "same_chunks_resident_frames_per_s": F * args.steps / dt_res,
           "pcie_inclusive_over_resident": dt_res / dt,
           "kernels_ms_per_chunk": kms}
    emit(out, args.detail)
    vb.close()
    return 0


# ------------------------------------------------------------------------------------------------
# sub-benchmarks of the default run: BASELINE configs 2, 3 and 4, a few steps each, outside the headline's timed region
# ------------------------------------------------------------------------------------------------
def sub_benchmarks(vb, torch, dev, pkg, audio48, F48):
    """audio48: the pipeline's resident recording (F48 frames at 1200 / 480).  Returns a list of compact bench records."""
    f64, i32 = torch.float64, torch.int32
    out = []
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])

    def record(name, wl, desc, F, frame_len, stride, step, steps, warmup, kmax=None, extra=None):
        r = guarded(name, lambda: _record(name, wl, desc, F, frame_len, stride, step, steps, warmup, kmax, extra), torch)
        out.append(r)

    def _record(name, wl, desc, F, frame_len, stride, step, steps, warmup, kmax, extra):
        dt, prof, work = timed(vb, torch, step, warmup, steps)
        roof, hbm, kms, kernels = roofline_for(wl, prof, work, F, frame_len, stride, steps)
        r = {"name": name, "workload": desc, "value": F * steps / dt, "unit": "frames/s", "steps": steps, "warmup": warmup,
             "ms_per_step": dt / steps * 1e3, "frames": F, "frame_len": frame_len, "hop": stride, "roofline": roof, "kernels_ms": kms}
        if kmax is not None:
            r["pitch_kmax"] = kmax
        if hbm is not None:
            r["roofline_hbm"] = {k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_frame")}
        if extra:
            r.update(extra(dt, kernels, steps))
        return r

    # ---- config 3: Boersma pitch path, 10 h (3,599,998 frames) of the resident recording ----------------------------
    win = vb.window(pkg.WINDOW_HANNING, N48)
    F3 = min(CONFIG3_FRAMES, F48)
    for kmax, Fk, steps in ((1, F3, 2), (8, min(F3, 720_000), 2), (pkg.pitch_max_candidates(N48), min(F3, 360_000), 1)):
        cand = torch.empty((Fk, kmax, 2), dtype=f64, device=dev)
        cnt = torch.empty(Fk, dtype=i32, device=dev)
        pst = torch.empty(Fk, dtype=i32, device=dev)

        def step3(i, kmax=kmax, Fk=Fk, cand=cand, cnt=cnt, pst=pst):
            vb.pitch(audio48, SR, 0.2, 75.0, 600.0, kmax=kmax, frame_len=N48, stride=H48, n_frames=Fk, window=win, out=(cand, cnt, pst))
        what = "PitchExtractor output" if kmax == 1 else f"first {kmax} of the sorted candidate Vec" if kmax == 8 else "the WHOLE candidate Vec"
        record(f"config3_kmax{kmax}", "config3", f"Boersma pitch path, {Fk} frames of the synthetic 48 kHz recording, 25 ms / 10 ms hop, "
               f"kmax = {kmax} ({what})", Fk, N48, H48, step3, steps, 1, kmax=kmax)
        del cand, cnt, pst

    # ---- configs 2 and 4: dense [1,000,000, 512] f64 frames -------------------------------------------------------
    Fd = 1_000_000
    try:
        dense = torch.empty(Fd * 512, dtype=f64, device=dev)
        vb.synth_speech(Fd * 512, sample_offset=0, sample_rate=SR, out=dense)
    except Exception as e:  # noqa: BLE001
        out.append({"name": "config2", "error": repr(e)[:300]})
        out.append({"name": "config4", "error": repr(e)[:300]})
        return out
    win512 = vb.window(pkg.WINDOW_HANNING, 512)
    o_r = torch.empty((Fd, P + 1), dtype=f64, device=dev)
    o_a = torch.empty((Fd, P + 1), dtype=f64, device=dev)
    record("config2", "config2", f"batched autocorrelation(13) + LPC order-12, {Fd} x 512-sample f64 frames",
           Fd, 512, 512, lambda i: vb.autocorr_lpc(dense, P, frame_len=512, stride=512, n_frames=Fd, window=win512, out=(o_r, o_a)), 5, 2)
    del o_r, o_a
    seg = np.arange(0, Fd, SEG_FRAMES, dtype=np.int64)
    ff = {"formants": torch.empty((Fd, 4, 2), dtype=f64, device=dev), "res": None, "count": None, "coeffs": None,
          "status": torch.empty(Fd, dtype=i32, device=dev)}

    def whole4(dt, kernels, steps):
        return {"whole_config": config4_whole(Fd, dt / steps, kernels, steps)}
    record("config4", "config4", f"LPC(Burg)->Laguerre roots->formant track, {Fd} x 512-sample f64 frames, utterances of {SEG_FRAMES}",
           Fd, 512, 512, lambda i: vb.find_formants(dense, SR, P, est0, seg_start=seg, frame_len=512, stride=512, n_frames=Fd, out=ff), 5, 2,
           extra=whole4)
    del dense, ff
    out.append(guarded("pipeline_shapes", lambda: {
        "name": "pipeline_shapes", "workload": "the full pipeline (pitch + LPC + formants + MFCC, one utterance) on 1 h of the same "
        "recording at other frame shapes: 25 ms / 10 ms at 16, 24, 32, 44.1 kHz-like sample counts, the reference's own "
        "1024 / 512 (tests/lib.rs:56-57) and 2048 / 1024 (examples/pitch_detection.rs:23), and 4096 / 2048 (benches/periodic.rs:22-25)",
        "unit": "frames/s", "steps": 2, "warmup": 1, "shapes": pipeline_shapes(vb, torch, dev, pkg, audio48)}, torch))
    out.append(guarded("speech_44k", lambda: bench_speech(vb, torch, dev, pkg), torch))
    torch.cuda.empty_cache()          # torch's cached blocks back to the driver before the 138 GB recording + the library's own hipMallocs
    out.append(guarded("config5_100h_1gpu", lambda: config5_whole_on_one_gpu(vb, torch, dev, pkg), torch))
    return out


SPEECH_WAV = os.path.join(ROOT, "tests", "golden", "sample-two_vowels.wav")
SPEECH_SHAPES = ((1103, 441), (1024, 512))


def bench_speech(vb, torch, dev, pkg, hours=1.0, steps=2, warmup=1, shapes=SPEECH_SHAPES, order=13):
    """The full pipeline on REAL speech: what the reference's callers feed it (tests/lib.rs:60-83,
    examples/formant_extraction/src/main.rs:36-47 read WAV files).  Per shape: frames/s on the speech recording and on the
    synthetic signal with the same parameters, and the share of frames the one-pass Burg's guard / the conjugate-pair root
    finder's check hand to the reference's own recursions (`burg_direct`, `roots_direct`): the fast paths' coverage is a
    property of the material, so it is measured on material."""
    import wave
    from importlib import import_module
    with wave.open(SPEECH_WAV, "rb") as w:
        sr = float(w.getframerate())
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    ns = int(hours * 3600 * sr)
    syn = import_module(pkg.__name__ + ".synth")
    speech = syn.speech_recording(torch, dev, pcm, ns)
    synth = torch.empty(ns, dtype=torch.float64, device=dev)
    vb.synth_speech(ns, sample_offset=0, sample_rate=sr, out=synth)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    params = pkg.AnalysisParams.make(sr, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=order, est_init=est0,
                                     mfcc=(13, 100.0, 8000.0))
    REC = int(vb.L.vbx_record_doubles(params))
    rows = []
    for n, hop in shapes:
        F = pkg.frame_count(ns, n, hop)
        rec = torch.empty((F, REC), dtype=torch.float64, device=dev)
        st3 = torch.empty((3, F), dtype=torch.int32, device=dev)
        row = {"frame_len": n, "hop": hop, "frames": F}
        for name, audio in (("speech", speech), ("synthetic", synth)):
            def step(i, audio=audio, n=n, hop=hop, F=F, rec=rec, st3=st3):
                vb.analyze_frames(audio, params, frame_len=n, stride=hop, n_frames=F, out=rec, record_ld=REC, status=st3)
            dt, prof, work = timed(vb, torch, step, warmup, steps)
            kms = {k: round(ms / max(c, 1), 3) for k, (ms, c) in prof.items()}
            frames_w, cand_w, evals_w, terms_w = work
            row[name] = {"value": F * steps / dt, "ms_per_step": dt / steps * 1e3,
                         # what the pitch refinement did (device counters): the material decides it -- a recording that is
                         # voiced throughout refines a candidate in (nearly) every frame, the synthetic one in four of five
                         "sinc_evals_per_frame": evals_w / max(frames_w, 1), "sinc_terms_per_frame": terms_w / max(frames_w, 1),
                         "candidates_per_frame": cand_w / max(frames_w, 1),
                         "burg_direct": vb.last_burg_direct_count() / F, "roots_direct": vb.last_roots_direct_count() / F,
                         "lpc_exact": vb.last_lpc_exact_count() / F,
                         "frames_with_nonzero_status": int((st3 != 0).any(dim=0).sum().item()),
                         "dominant_kernel": dominant_kernel(prof), "kernels_ms": kms}
        row["speech_over_synthetic"] = row["speech"]["value"] / row["synthetic"]["value"]
        rows.append(row)
        del rec, st3
    del speech, synth
    return {"name": "speech_44k", "workload": f"the full pipeline (pitch + LPC + formants order {order} + MFCC, one utterance) on "
            f"{hours:g} h of real speech at {sr:g} Hz (tests/golden/sample-two_vowels.wav tiled: per-tile gain in [0.5, 1), -70 dB "
            "dither) and on the synthetic signal with the same parameters", "unit": "frames/s", "steps": steps, "warmup": warmup,
            "sample_rate": sr, "formant_order": order, "shapes": rows}


def config5_whole_on_one_gpu(vb, torch, dev, pkg, hours=100.0, steps=3, warmup=1):
    """BASELINE config 5 WHOLE on one GPU: 100 h = 36,000,000 frames, 138 GB of f64 audio resident beside the headline's shard
    (288 GB of HBM hold both), ONE utterance, one vbx_analyze_frames_f64 call per step.  Skipped (and says so) if the
    allocation fails."""
    F = int(hours * 3600 * 100)
    ns = (F - 1) * H48 + N48
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0, mfcc=(13, 100.0, 8000.0))
    REC = int(vb.L.vbx_record_doubles(params))
    name = "config5_100h_1gpu"
    try:
        audio = torch.empty(ns, dtype=torch.float64, device=dev)
        rec = torch.empty((F, REC), dtype=torch.float64, device=dev)
        st3 = torch.empty((3, F), dtype=torch.int32, device=dev)
    except RuntimeError as e:                                   # out of memory on a smaller part: not a measurement
        return {"name": name, "skipped": repr(e)[:200]}
    vb.synth_speech(ns, sample_offset=0, sample_rate=SR, out=audio)

    def step(i):
        vb.analyze_frames(audio, params, frame_len=N48, stride=H48, n_frames=F, out=rec, record_ld=REC, status=st3)
    dt, prof, work = timed(vb, torch, step, warmup, steps)
    roof, hbm, kms, _ = roofline_for("pipeline", prof, work, F, N48, H48, steps)
    bad = int((st3 != 0).sum().item())
    r = {"name": name, "workload": f"BASELINE config 5 whole on ONE GPU: {hours:g} h synthetic 48 kHz = {F} frames, 25 ms / 10 ms hop, "
         "pitch + LPC + formants + MFCC, the whole recording one utterance, audio resident (138 GB)",
         "value": F * steps / dt, "unit": "frames/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "frames": F,
         "frame_len": N48, "hop": H48, "frames_with_nonzero_status": bad, "roofline": roof, "kernels_ms": kms}
    if hbm is not None:
        r["roofline_hbm"] = {k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_frame")}
    del audio, rec, st3
    return r


PIPELINE_SHAPES = ((400, 160), (512, 256), (600, 240), (800, 320), (1024, 512), (1102, 441), (1103, 441), (1200, 480), (1600, 640),
                   (2048, 1024), (3000, 1200), (4000, 2000), (4096, 2048),
                   (4096, 1024))       # the last one: benches/periodic.rs:29-39 (bin 4096, hop 1024)


def pipeline_shapes(vb, torch, dev, pkg, audio48, hours=1.0, shapes=PIPELINE_SHAPES):
    """vbx_analyze_frames_f64 (everything on) over `hours` of the resident recording viewed at each frame shape: frames/s,
    per-kernel ms of one step and the dominant kernel.  A few steps each: driver-timed numbers for the shapes off the headline."""
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0, mfcc=(13, 100.0, 8000.0))
    REC = int(vb.L.vbx_record_doubles(params))
    ns = min(int(hours * 3600 * SR), int(audio48.numel()))
    rows = []
    for n, hop in shapes:
        F = pkg.frame_count(ns, n, hop)
        rec = torch.empty((F, REC), dtype=torch.float64, device=dev)
        st3 = torch.empty((3, F), dtype=torch.int32, device=dev)

        def step(i, n=n, hop=hop, F=F, rec=rec, st3=st3):
            vb.analyze_frames(audio48, params, frame_len=n, stride=hop, n_frames=F, out=rec, record_ld=REC, status=st3)
        dt, prof, _ = timed(vb, torch, step, 1, 2)
        kms = {k: round(ms / max(c, 1), 3) for k, (ms, c) in prof.items()}
        dom = dominant_kernel(prof)
        rows.append({"frame_len": n, "hop": hop, "frames": F, "value": F * 2 / dt, "ms_per_step": dt / 2 * 1e3, "dominant_kernel": dom,
                     "kernels_ms": kms, "beside_it": sorted(k for k in kms if PROF_STREAMS.get(k, 0) != 0),
                     # the fast paths' hand-overs (shares of the frames): one-pass Burg -> direct recursion, conjugate-pair roots ->
                     # the reference's iteration, Levinson rows redone in double-double
                     "burg_direct": vb.last_burg_direct_count() / F, "roots_direct": vb.last_roots_direct_count() / F,
                     "lpc_exact": vb.last_lpc_exact_count() / F})
        del rec, st3
    return rows


def cross_rank_check(vb, torch, dev, pkg, params, REC, rows_all, frame_len, stride, cuts, reach=4096, settle=1024):
    """Rank 0, after the timed steps, outside the timed region: the frames on either side of every shard cut, analysed AGAIN on
    this GPU alone as one stretch of the recording, against the rows the ranks produced and the gather delivered.  The stretch
    starts `reach` frames before the cut; its own tracker has forgotten its start long before `settle` frames (64 suffice
    almost always), so rows [cut - reach + settle, cut + reach) must be BIT FOR BIT the gathered ones -- every column: pitch,
    LPC, MFCC (per-frame) and the formant tracks (carried across the cut by warm-up + state hand-off).  Returns counts and a
    verdict; the caller prints the line and then FAILS the run (exit code 3, `valid: false`) on anything but "bit-identical":
    a headline number over rows that are not the single-GPU rows is not a measurement of this path."""
    out = {"cuts": [int(c) for c in cuts], "rows_compared": 0, "rows_different": 0}
    try:
        total = int(rows_all.shape[0])
        for c in cuts:
            lo, hi = max(0, int(c) - reach), min(total, int(c) + reach)
            ns = (hi - lo - 1) * stride + frame_len
            a = torch.empty(ns, dtype=torch.float64, device=dev)
            vb.synth_speech(ns, sample_offset=lo * stride, sample_rate=SR, out=a)
            r = torch.empty((hi - lo, REC), dtype=torch.float64, device=dev)
            st = torch.empty((3, hi - lo), dtype=torch.int32, device=dev)
            vb.analyze_frames(a, params, frame_len=frame_len, stride=stride, n_frames=hi - lo, out=r, record_ld=REC, status=st)
            vb.sync()
            first = 0 if lo == 0 else settle
            mine, theirs = r[first:], rows_all[lo + first:hi]
            # bit patterns, not values: NaN == NaN, -0.0 != 0.0
            diff = (mine.view(torch.int64) != theirs.view(torch.int64)).any(dim=1)
            out["rows_compared"] += int(diff.numel())
            out["rows_different"] += int(diff.sum().item())
            del a, r, st
        out["verdict"] = "bit-identical" if out["rows_different"] == 0 else "DIFFERENT"
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)[:300]
    return out


# ------------------------------------------------------------------------------------------------
# the ONE stdout line: contract keys + config + roofline + roofline_hbm + kernels_ms + cpu_baseline + {sub-benchmark: frames/s}.
# Everything else (tables, prose) goes to the --detail file.  Round 5's line grew to 25 KB and the driver, which keeps the last
# 8 KB of stdout, could not parse it: LINE_LIMIT is asserted here, in tests/test_bench_contract.py and in tests/test_gpu_bench_line.py.
# ------------------------------------------------------------------------------------------------
LINE_LIMIT = 6000
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "issue_frac", "valu_insts_per_frame", "ms_avg", "frames_per_launch",
                 "launches_per_step", "algorithmic_bytes_per_frame", "flops_per_frame", "sinc_terms_per_frame")


def _json_default(o):
    """numpy scalars / arrays that found their way into a record must not cost the line"""
    if isinstance(o, np.generic):
        return o.item()
    if isinstance(o, np.ndarray):
        return o.tolist()
    return str(o)


def _sig(x, digits=6):
    """Floats to `digits` significant digits (the detail file keeps them whole); containers recursively."""
    if isinstance(x, np.generic):
        x = x.item()
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _slim_roofline(r):
    if r is None:
        return None
    s = {k: r[k] for k in ROOFLINE_KEYS if k in r}
    src = r.get("traffic_source")
    if isinstance(src, dict):
        s["traffic_bytes_per_frame"] = src.get("bytes_per_frame")
        s["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate child passes"
                               if str(src.get("source", "")).startswith("live") else
                               "committed PMC pass: profiles/pmc_traffic.json @ " + str(src.get("commit")))
    ref = r.get("reference_sums_at_peak")
    if isinstance(ref, dict):
        s["reference_sums_at_peak_ratio"] = ref.get("ratio")
    return s


def _sub_value(s):
    """One number (or a small map of numbers) per sub-benchmark."""
    if "error" in s or "skipped" in s:
        return {"error": str(s.get("error", s.get("skipped")))[:120]}
    if "value" in s:
        return s["value"]
    if s.get("name") == "speech_44k":
        return {f"{r['frame_len']}/{r['hop']}": {"speech": r["speech"]["value"], "synthetic": r["synthetic"]["value"],
                                                 "burg_direct": r["speech"]["burg_direct"]} for r in s.get("shapes", [])}
    if "shapes" in s:
        return {f"{r['frame_len']}/{r['hop']}": r["value"] for r in s["shapes"]}
    return None


def compact_line(out, detail_path=None):
    """The driver's line from the full record `out`."""
    line = {k: out[k] for k in CONTRACT_KEYS if k in out}
    cfg = dict(out.get("config") or {})
    for k in ("gather", "parallelism", "tracker_across_ranks"):           # prose: shortened, whole in the detail file
        if isinstance(cfg.get(k), str) and len(cfg[k]) > 90:
            cfg[k] = cfg[k][:87] + "..."
    line["config"] = cfg
    if "roofline" in out:
        line["roofline"] = _slim_roofline(out["roofline"])
    if out.get("roofline_hbm") is not None:
        line["roofline_hbm"] = _slim_roofline(out["roofline_hbm"])
    if "kernels_ms" in out:
        line["kernels_ms"] = out["kernels_ms"]
    c = out.get("cpu_baseline")
    if c is not None:
        c = dict(c)
        if isinstance(c.get("sample"), str) and len(c["sample"]) > 200:
            c["sample"] = c["sample"][:197] + "..."
        line["cpu_baseline"] = c
    for k in ("valid", "invalid_because", "live_traffic_error"):
        if k in out:
            line[k] = str(out[k])[:160] if k == "live_traffic_error" else out[k]
    if "cross_rank_check" in out:
        line["cross_rank_check"] = {k: out["cross_rank_check"].get(k) for k in ("verdict", "rows_compared", "rows_different", "error")
                                    if k in out["cross_rank_check"]}
    if "whole_config" in out:
        w = out["whole_config"]
        line["whole_config"] = {k: w[k] for k in ("bytes_per_frame", "GBps", "hbm_frac", "flops_per_frame", "fp64_frac") if k in w}
    if "sub_benchmarks" in out:
        line["sub_benchmarks"] = {s.get("name", f"sub{i}"): _sub_value(s) for i, s in enumerate(out["sub_benchmarks"])}
    if "speech" in out:
        line["speech"] = _sub_value(out["speech"])
    if "parity" in out:
        line["parity"] = "GPU == oracle through the C ABI (tests/ -m gpu); oracle pinned by the reference's 26 KATs + WAV fixtures; see detail"
    if detail_path:
        line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path
    text = json.dumps(_sig(line), default=_json_default)
    if len(text) > LINE_LIMIT:                                           # never again an unparsable line: shed the optional parts
        for k in ("sub_benchmarks", "parity", "kernels_ms", "roofline_hbm"):
            if k == "sub_benchmarks" and isinstance(line.get(k), dict):
                line[k] = {n: (v if isinstance(v, (int, float)) else "see detail") for n, v in line[k].items()}
            else:
                line.pop(k, None)
            text = json.dumps(_sig(line), default=_json_default)
            if len(text) <= LINE_LIMIT:
                break
    return text


def emit(out, detail_path):
    """Write the full record to the detail file (best effort) and print the compact line."""
    written = None
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1, default=_json_default)
                f.write("\n")
            written = os.path.abspath(detail_path)
        except OSError as e:
            sys.stderr.write(f"bench.py: could not write {detail_path}: {e}\n")
    print(compact_line(out, written), flush=True)


def guarded(name, fn, torch=None):
    """A sub-benchmark must never lose the headline that was already measured: any exception becomes a record."""
    try:
        return fn()
    except Exception as e:  # noqa: BLE001
        sys.stderr.write(f"bench.py: sub-benchmark {name} failed: {e!r}\n")
        if torch is not None:
            try:
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            except Exception:  # noqa: BLE001
                pass
        return {"name": name, "error": repr(e)[:300]}


# ------------------------------------------------------------------------------------------------
# the pipeline's step at any world size.  tests/test_shard_cpu.py runs exactly these two functions at world 8 over gloo with test
# doubles for the context and the communicator (the N > 1 path has never run on hardware: its control flow at least has run).
# ------------------------------------------------------------------------------------------------
def pipeline_buffers(torch, dev, world, rank, F, FA, REC):
    """The record buffers of one rank: `rec[b]` is what analyze writes ([FA, REC]: the rank's warm-up rows, then its own F rows),
    `gathered[b]` what the gather fills on rank 0 ([world * F, REC]; None elsewhere).  Rank 0 has no warm-up rows and writes its
    rows in place, straight into the gathered array.  Double-buffered at N > 1: step i + 1's kernels overlap step i's transfer."""
    f64 = torch.float64
    nbuf = 2 if world > 1 else 1
    if rank == 0:
        gathered = [torch.empty((world * F, REC), dtype=f64, device=dev) for _ in range(nbuf)]
        rec = [g[:F] for g in gathered]                                   # rank 0 owns rows [0, F): written in place
    else:
        gathered = [None] * nbuf
        rec = [torch.empty((FA, REC), dtype=f64, device=dev) for _ in range(nbuf)]      # rows [warm, FA) are the rank's own
    return rec, gathered


def pipeline_step(vb, comm, audio, params, seg, frame_len, stride, FA, warm, REC, rec, gathered, st3, counts, plan, stitch):
    """step(i): wait for buffer i % nbuf's last transfer (device-side) -> analyze -> tracker hand-off along the ranks (an
    utterance cut by the rank boundary) -> gather of the rank's OWN rows (past the warm-up rows) to rank 0."""
    def step(i):
        b = i % len(rec)
        if comm is not None:
            comm.wait(b)                                                  # device-side: buffer b's last transfer is done
        vb.analyze_frames(audio, params, seg_start=seg, frame_len=frame_len, stride=stride, n_frames=FA,
                          out=rec[b], record_ld=REC, status=st3)
        if stitch:                 # the tracker's state along the chain of ranks; formant columns start at double 2 of a record
            comm.stitch_tracks(rec[b].data_ptr() + 16, FA, REC, plan, None, slot=b)
        if comm is not None:       # per-frame records to rank 0 over RCCL/xGMI
            comm.gather_records(rec[b].data_ptr() + warm * REC * 8, counts, REC, 0, out=gathered[b], slot=b)
    return step


# ------------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dry = bool(os.environ.get("VBX_BENCH_DRY_RUN"))      # tests/test_shard_cpu.py: the launcher's plumbing without a GPU
    rc_final = 0                                         # 3: the line was printed but its cross-rank check failed
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # control plane: gloo on CPU tensors.  torch never brings up RCCL in this process; the library's communicator
        # (below) is the only RCCL instance.
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    pkg = graft.load_package()
    N, H = args.frame_len, args.hop
    wl = args.workload
    # shard geometry (host arithmetic only: also what the dry run reports)
    plan = None
    if wl in ("pipeline", "config3"):
        total_frames_per_gpu = int(round(args.hours * 3600 * SR / H))         # 100 frames per second at the default hop
        total_frames_per_gpu -= total_frames_per_gpu % SEG_FRAMES
        F = max(total_frames_per_gpu, SEG_FRAMES)
        # weak scaling: the recording is world * F frames long and splits by contiguous frame ranges.  One utterance (the
        # default) or utterances of --utterance-frames frames; a cut inside an utterance is stitched (vbx_shard_plan).
        UTT = args.utterance_frames
        seg_all = None if UTT <= 0 else np.arange(0, world * F, UTT, dtype=np.int64)
        plan = pkg.shard_plan(world * F, world, rank, seg_all)
        lo, hi, warm = plan.lo, plan.hi, plan.warm
        if hi - lo != F:         # a cut that moved to a nearby utterance start (vbx_shard_range): shards of unequal length
            raise SystemExit(f"bench.py: --utterance-frames {UTT} moves the shard cuts ({hi - lo} frames on rank {rank}, not {F}); "
                             "weak scaling wants equal shards: pass a length that divides the per-GPU frame count, or 0")
        seg = pkg.shard_local_segments(plan, seg_all)                         # utterance starts of frames [lo - warm, hi)
        s0, s1 = pkg.shard_samples(lo - warm, hi, N, H)                       # includes the frame_len - hop halo
    else:
        F = args.frames
        warm = 0
        seg = np.arange(0, F, SEG_FRAMES, dtype=np.int64)
    counts = [F] * world
    gather_desc = ("library (vbx_gather_records_f64: grouped ncclSend/ncclRecv on the communicator's own stream, one direct "
                   "xGMI link per peer; control plane = torch.distributed gloo)") if world > 1 else None
    if dry:
        REC = 36
        off, cnt, op = pkg.gather_plan(counts, rank, 0, REC)
        seen = [None] * world
        mine = (rank, local, os.environ.get("MASTER_PORT"), [int(x) for x in op], [int(x) for x in off],
                plan.as_dict() if plan is not None else None)
        if world > 1:
            dist.all_gather_object(seen, mine)
        else:
            seen = [mine]
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": [list(s[:3]) for s in seen], "gpus_arg": args.gpus,
                              "control_plane": dist.get_backend() if world > 1 else None, "data_plane": gather_desc,
                              "rccl_comms_per_rank_planned": 1 if world > 1 else 0, "torch_nccl_process_groups": 0,
                              "gather_plan": {"rows": counts, "record_doubles": REC, "ops_by_rank": [s[3] for s in seen],
                                              "offsets": seen[0][4]},
                              "utterance_frames": args.utterance_frames, "shard_plans": [s[5] for s in seen]}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0
    if os.environ.get("VBX_BENCH_ONE_GPU"):     # test hook: every rank on GPU 0 (whether RCCL accepts that is up to RCCL)
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the product has no CPU path")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # one explicit HIP stream shared by torch (allocation) and the library's kernels; the library adds its own side
    # stream (formant chain beside the pitch kernel) and the communicator's transfer stream
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    vb = pkg.VoxBox(local, tstream.cuda_stream)

    f64 = torch.float64
    if wl == "frontend":
        return bench_frontend(args, torch, dev, vb, pkg)
    if args.signal == "speech":
        if wl != "pipeline" or world != 1:
            raise SystemExit("bench.py --signal speech: the pipeline on one GPU only")
        r = bench_speech(vb, torch, dev, pkg, hours=min(args.hours, 2.0) if args.hours != 12.5 else 1.0, steps=args.steps, warmup=args.warmup)
        head = r["shapes"][0]
        emit({"metric": "frames/sec (pipeline on real 44.1 kHz speech, 1103-sample frames / 441-sample hop, order 13)",
              "value": head["speech"]["value"], "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": head["speech"]["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "f64", "data": "real speech (committed WAV fixture, tiled)", "config": {"workload": r["workload"]},
              "speech": r}, args.detail)
        vb.close()
        return 0
    if args.host_fed:
        if wl != "pipeline" or world != 1 or (N, H) != (N48, H48):
            raise SystemExit("bench.py --host-fed: the default pipeline shape on one GPU only")
        return bench_host_fed(args, torch, dev, vb, pkg)
    comm = None
    if world > 1:
        # the library's RCCL communicator for the record gather; the 128-byte id travels over the gloo control plane.
        # Every rank learns whether every other rank's communicator came up; if not, the bench FAILS: there is no other
        # transport for the records.
        ids = [None]
        err = ""
        if rank == 0:
            try:
                ids = [pkg.comm_unique_id()]
            except pkg.VoxBoxError as e:
                err = str(e)
        dist.broadcast_object_list(ids, src=0)
        ok = 0
        if ids[0] is not None:
            try:
                comm = pkg.Comm(vb, ids[0], world, rank)
                ok = 1
            except pkg.VoxBoxError as e:
                err = str(e)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm is not None:
                comm.close()
            raise SystemExit(f"bench.py rank {rank}: the library's RCCL communicator did not come up on every rank "
                             f"({err or 'another rank failed'}); the record gather has no fallback transport")
        assert pkg.comm_live_count() == 1

    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    if wl in ("pipeline", "config3"):
        audio = torch.empty(s1 - s0, dtype=f64, device=dev)
        vb.synth_speech(s1 - s0, sample_offset=s0, sample_rate=SR, out=audio)
        frame_len, stride = N, H
        win = vb.window(pkg.WINDOW_HANNING, N)
        desc = (f"{'full pitch+LPC+formants+MFCC pipeline' if wl == 'pipeline' else 'Boersma pitch path'}, "
                f"{args.hours:g} h/GPU synthetic 48 kHz, " +
                ("25 ms / 10 ms hop" if (N, H) == (N48, H48) else f"{N}-sample frames / {H}-sample hop"))
    else:
        audio = torch.empty(F * 512, dtype=f64, device=dev)
        vb.synth_speech(F * 512, sample_offset=rank * F * 512, sample_rate=SR, out=audio)
        frame_len, stride = 512, 512
        win = vb.window(pkg.WINDOW_HANNING, 512)
        desc = ("batched autocorrelation + LPC order-12" if wl == "config2" else
                "LPC(Burg)->Laguerre roots->formant track") + f", {F} x 512-sample f64 frames/GPU"
    FA = F + warm                 # frames this rank analyses: its range plus the tracker's warm-up frames before it

    # outputs (torch owns the device memory; the C ABI gets raw pointers).  The pipeline writes one record per frame
    # straight into the buffer the gather sends (rank 0: straight into the gathered array), double-buffered so that
    # step i+1's kernels overlap step i's transfer.
    if wl == "pipeline":
        params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0,
                                         mfcc=(13, 100.0, 8000.0))
        REC = int(vb.L.vbx_record_doubles(params))
        rec, gathered = pipeline_buffers(torch, dev, world, rank, F, FA, REC)
        st3 = torch.empty((3, FA), dtype=torch.int32, device=dev)
        stitch = comm is not None and (plan.continues_prev or plan.continues_next)
        step = pipeline_step(vb, comm, audio, params, seg, frame_len, stride, FA, warm, REC, rec, gathered, st3, counts, plan, stitch)
    else:
        REC = 0
        o_cand = torch.empty((F, args.kmax, 2), dtype=f64, device=dev)
        o_cnt = torch.empty(F, dtype=torch.int32, device=dev)
        o_pst = torch.empty(F, dtype=torch.int32, device=dev)
        o_r = torch.empty((F, P + 1), dtype=f64, device=dev)
        o_a = torch.empty((F, P + 1), dtype=f64, device=dev)
        o_form = torch.empty((F, 4, 2), dtype=f64, device=dev)
        o_fst = torch.empty(F, dtype=torch.int32, device=dev)
        ff = {"formants": o_form, "res": None, "count": None, "coeffs": None, "status": o_fst}

        def step(i):
            if wl == "config4":
                vb.find_formants(audio, SR, P, est0, seg_start=seg, frame_len=frame_len, stride=stride, n_frames=F, out=ff)
            elif wl == "config2":
                vb.autocorr_lpc(audio, P, frame_len=frame_len, stride=stride, n_frames=F, window=win, out=(o_r, o_a))
            elif wl == "config3":
                vb.pitch(audio.data_ptr() + warm * stride * 8, SR, 0.2, 75.0, 600.0, kmax=args.kmax, frame_len=frame_len, stride=stride,
                         n_frames=F, window=win, out=(o_cand, o_cnt, o_pst))

    # N > 1 has never run on hardware in this build's rounds: a collective that never completes must not hang the node.  A
    # watchdog ends THIS rank (exit code 5, a message on stderr) if warm-up + the timed steps take absurdly long; the launcher
    # (or torchrun) then ends the others.
    watchdog = None
    if world > 1:
        import threading
        limit = float(os.environ.get("VBX_BENCH_WATCHDOG_S", "0")) or max(300.0, 30.0 * (args.warmup + args.steps) * args.hours / 12.5)

        def _bark():
            sys.stderr.write(f"bench.py rank {rank}: warm-up + {args.steps} steps did not finish within {limit:.0f} s -- a collective of the "
                             "N > 1 path (tracker hand-off chain or record gather) is stuck; aborting this rank\n")
            sys.stderr.flush()
            os._exit(5)
        watchdog = threading.Timer(limit, _bark)
        watchdog.daemon = True
        watchdog.start()
    dt, prof, work = timed(vb, torch, step, args.warmup, args.steps, barrier if world > 1 else None)
    if watchdog is not None:
        watchdog.cancel()
    if world > 1:
        tt = torch.tensor([dt], dtype=f64)                                   # CPU tensor: gloo
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total = F * world * args.steps
        default_shape = wl not in ("pipeline", "config3") or (frame_len, stride) == (N48, H48)
        roof, hbm, kms, kernels = roofline_for(wl, prof, work, F, frame_len, stride, args.steps)
        out = {
            "metric": METRIC if wl == "pipeline" and default_shape else
                      f"frames/sec ({wl})" if default_shape else f"frames/sec ({wl}, {frame_len}-sample frames / {stride}-sample hop)",
            "value": total / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "frames_per_gpu": F, "frame_len": frame_len, "hop": stride,
                       "lpc_order": P, "mfcc": 13, "pitch_kmax": args.kmax if wl != "pipeline" else 1,
                       "utterance_frames": (args.utterance_frames if args.utterance_frames > 0 else "whole recording = one utterance")
                                           if wl == "pipeline" else None,
                       "tracker_across_ranks": ("warm-up of %d frames + state hand-off along the ranks (vbx_comm_stitch_tracks_f64)" % warm
                                                if (world > 1 and plan is not None and (plan.continues_prev or plan.continues_next)) else None),
                       "record_bytes": REC * 8,
                       "parallelism": f"frame-range split x{world}, one process per GPU, RCCL gather of the records to rank 0",
                       "gather": gather_desc, "control_plane": "gloo (CPU tensors)" if world > 1 else None,
                       "rccl_comms_per_rank": pkg.comm_live_count(), "torch_nccl_process_groups": 0},
            "roofline": roof,
            "kernels_ms": kms,
        }
        if hbm is not None:
            out["roofline_hbm"] = hbm
        # the headline kernel's HBM traffic measured in THIS run (two short rocprofv3 --pmc child passes); on any failure the committed
        # evidence file's figure stays, and traffic_source says which it is
        if (wl == "pipeline" and default_shape and world == 1 and not args.no_live_traffic and roof.get("kernel") in LIVE_KERNELS
                and not os.environ.get("ROCP_TOOL_LIBRARIES") and "rocprof" not in os.environ.get("LD_PRELOAD", "")):      # (not under a profiler already)
            lt = guarded("live_traffic", lambda: live_traffic(roof["kernel"]))
            if isinstance(lt, tuple):
                bpf, src = lt
                for r in (roof, hbm):
                    if r is not None:
                        r["traffic_committed_evidence"] = r.get("traffic_source")
                        r["traffic"] = bpf * r["frames_per_launch"]
                        r["traffic_source"] = src
                if "issue_frac" in roof:                              # the vector ALU's busy share and the instruction count, from this run's SQ pass
                    roof["issue_frac_committed_evidence"] = roof["issue_frac"]
                    roof["issue_frac"], roof["valu_insts_per_frame"] = src["valu_busy"], src["valu_insts_per_frame"]
                    roof["issue_frac_source"] = "live: rocprofv3 --pmc SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 / 4 x 1024 SIMDs), a child pass of this bench"
            else:
                out["live_traffic_error"] = lt.get("error")
        # what "parity" means for this line: GPU == oracle is tested; oracle == reference is pinned
        # by the reference's own known-answer tests where it has any, and is NOT where it has none
        out["parity"] = PARITY_NOTE
        if wl == "config4":
            # three kernels in sequence -- Burg, the root finder, the chunked tracker scan (four launches + a sweep, timed as
            # one) -- so besides the dominant kernel's line: the whole config against both roofs (4264 B and
            # ~135 kflop per frame)
            out["whole_config"] = config4_whole(F, dt / args.steps, kernels, args.steps)
        # N > 1 (or VBX_BENCH_SELFCHECK=1 at N = 1, a rehearsal of the same code against an interior "cut"): are the gathered rows
        # around every shard cut what ONE GPU computes for the same stretch of the recording?
        if wl == "pipeline" and args.utterance_frames <= 0 and (world > 1 or os.environ.get("VBX_BENCH_SELFCHECK")):
            vb.sync()
            if comm is not None:
                comm.sync()
            b_last = (args.warmup + args.steps - 1) % len(rec)
            rows_all = gathered[b_last] if world > 1 else rec[b_last]
            cuts = [r * F for r in range(1, world)] if world > 1 else [F // 2]
            chk = cross_rank_check(vb, torch, dev, pkg, params, REC, rows_all, frame_len, stride, cuts)
            out["cross_rank_check"] = chk
            if chk.get("verdict") != "bit-identical" or "error" in chk:
                out["valid"] = False
                out["invalid_because"] = "cross_rank_check: the gathered rows around a shard cut are not the single-GPU rows"
                rc_final = 3
        if not args.no_cpu:          # rank 0, outside the timed region, at every N
            out["cpu_baseline"] = guarded("cpu_baseline", lambda: cpu_baseline(wl, args.cpu_seconds, frame_len, stride))
        if wl == "pipeline" and default_shape and world == 1 and not args.no_sub:
            del rec, gathered                                                 # the records' HBM back before the dense batches
            out["sub_benchmarks"] = guarded("sub_benchmarks", lambda: sub_benchmarks(vb, torch, dev, pkg, audio, F), torch)
            if isinstance(out["sub_benchmarks"], dict):                       # the guard's error record
                out["sub_benchmarks"] = [out["sub_benchmarks"]]
        emit(out, args.detail)
    if comm is not None:
        comm.close()
    vb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc_final


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main() or 0)
