#!/usr/bin/env python3
"""Round 6: the Levinson conditioning probe + double-double redo (k_lpc_exact.hip) against the long-double arbiter, on EVERY row.
  python3 tools/experiments/lpc_exact_check.py [frames]
Per case (speech 1103/441 and 1024/512 at orders 12 -- the fused kernel's rows -- and 13 -- autocorrelate + levinson_rows --, the
synthetic shard at 1200/480 order 12, dense 512-sample frames order 12 = config 2): rows listed by the probe, worst distance of the
GPU's and of the oracle's rows from the long-double recursion on long-double lag sums (parity metric), rows beyond 1e-6 of the oracle,
with the probe on and off (VBX_LPC_EXACT=0); then timings of the headline call and of config 2 on / off."""
import json, os, sys, time, wave
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); o = g.load_oracle()
import importlib
syn = importlib.import_module(g.PKG_NAME + ".synth")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 6000


def metric(v, al):
    al = np.asarray(al, dtype=np.longdouble)
    return np.max(np.abs(v - al) / np.maximum(np.abs(al), 1e-6 * np.max(np.abs(al), axis=1, keepdims=True)), axis=1).astype(np.float64)


def arbiter(frames_w, p):
    """src/spectrum.rs:63-84 in long double on long-double lag sums (src/periodic.rs:284: seeded with x[0]), all rows at once"""
    xl = frames_w.astype(np.longdouble)
    n = xl.shape[1]
    r = np.stack([xl[:, 0] + np.sum(xl[:, 1:n - k] * xl[:, 1 + k:n], axis=1) for k in range(p + 1)], axis=1)
    a = np.zeros_like(r); a[:, 0] = 1; err = r[:, 0].copy()
    for i in range(1, p + 1):
        acc = r[:, i].copy()
        for j in range(1, i):
            acc = acc + a[:, j] * r[:, i - j]
        k = -acc / err
        t = a.copy(); a[:, i] = k
        for j in range(1, i):
            a[:, j] = t[:, j] + k * t[:, i - j]
        err = err * (1 - k * k)
    return a


def oracle_rows(frames_w, p):
    return np.stack([o.lpc(o.autocorrelate(fw, p + 1), p) for fw in frames_w])


def case(name, audio, sr, n, hop, p, F, dense=False):
    out = {"case": name, "frames": F, "frame_len": n, "hop": hop, "order": p}
    win = o.window("hanning", n)
    idx = np.arange(F)[:, None] * hop + np.arange(n)[None, :]
    fw = audio[idx] * win[None, :]
    al = arbiter(fw, p)
    ex = oracle_rows(fw, p)
    out["oracle_worst_vs_long_double"] = float(metric(ex, al).max())
    for mode in ("1", "0"):
        os.environ["VBX_LPC_EXACT"] = mode
        with pkg.VoxBox(0) as vb:
            ad = vb.to_device(audio)
            if dense:
                han = vb.window(pkg.WINDOW_HANNING, n)
                _, a = vb.autocorr_lpc(ad, p, frame_len=n, stride=hop, n_frames=F, window=han)
                listed = vb.last_lpc_exact_count()
            else:
                prm = pkg.AnalysisParams.make(sr, pitch=(0.2, 75.0, 600.0), lpc_order=p, formant_order=0, mfcc=(13, 100.0, 8000.0))
                rec, st = vb.analyze_frames(ad, prm, frame_len=n, stride=hop, n_frames=F)
                l0, ln = prm.columns()["lpc"]
                a = rec[:, l0:l0 + ln]
                listed = vb.last_lpc_exact_count()
        dg = metric(a, al)
        vs_o = np.max(np.abs(a - ex) / np.maximum(np.abs(ex), 1e-6 * np.max(np.abs(ex), axis=1, keepdims=True)), axis=1)
        out["probe_on" if mode == "1" else "probe_off"] = {
            "listed": listed, "gpu_worst_vs_long_double": float(dg.max()), "rows_gpu_beyond_1e-6_of_long_double": int((dg > 1e-6).sum()),
            "rows_beyond_1e-6_of_oracle": int((vs_o > 1e-6).sum()), "worst_vs_oracle": float(vs_o.max())}
    os.environ["VBX_LPC_EXACT"] = "1"
    print(json.dumps(out), flush=True)
    return out


def timing():
    res = {}
    for mode in ("1", "0", "1", "0"):
        os.environ["VBX_LPC_EXACT"] = mode
        with pkg.VoxBox(0) as vb:
            N, H, SR, Fh = 1200, 480, 48000.0, 720_000
            ad = vb.synth_speech((Fh - 1) * H + N)
            prm = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=12, formant_order=0, mfcc=(13, 100.0, 8000.0))
            REC = int(vb.L.vbx_record_doubles(prm))
            rec = vb.empty((Fh, REC)); st = vb.empty((3, Fh), np.int32)
            for _ in range(2):
                vb.analyze_frames(ad, prm, frame_len=N, stride=H, n_frames=Fh, out=rec, record_ld=REC, status=st)
            vb.sync(); t0 = time.perf_counter()
            for _ in range(5):
                vb.analyze_frames(ad, prm, frame_len=N, stride=H, n_frames=Fh, out=rec, record_ld=REC, status=st)
            vb.sync(); ta = (time.perf_counter() - t0) / 5
            listed = vb.last_lpc_exact_count()
            ad.free()
            Fd = 1_000_000
            dd = vb.synth_speech(Fd * 512); han = vb.window(pkg.WINDOW_HANNING, 512)
            o_r = vb.empty((Fd, 13)); o_a = vb.empty((Fd, 13))
            for _ in range(2):
                vb.autocorr_lpc(dd, 12, frame_len=512, stride=512, n_frames=Fd, window=han, out=(o_r, o_a))
            vb.sync(); t0 = time.perf_counter()
            for _ in range(10):
                vb.autocorr_lpc(dd, 12, frame_len=512, stride=512, n_frames=Fd, window=han, out=(o_r, o_a))
            vb.sync(); tc = (time.perf_counter() - t0) / 10
            l2 = vb.last_lpc_exact_count()
        res.setdefault("probe_" + ("on" if mode == "1" else "off"), []).append(
            {"analyze_720k_ms": ta * 1e3, "listed": listed, "config2_1M_ms": tc * 1e3, "config2_listed": l2})
    print(json.dumps({"timing": res}), flush=True)


with wave.open(os.path.join(ROOT, "tests", "golden", "sample-two_vowels.wav"), "rb") as w:
    sr = float(w.getframerate()); pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
import torch
for n, hop in ((1103, 441), (1024, 512)):
    speech = syn.speech_recording(torch, "cpu", pcm, (F - 1) * hop + n).numpy()
    for p in (12, 13):
        case(f"speech_{n}_{hop}_order{p}", speech, sr, n, hop, p, F)
synth = syn.synth_speech((F - 1) * 480 + 1200, sample_offset=0)
case("synthetic_1200_480_order12", synth, 48000.0, 1200, 480, 12, F)
dense = syn.synth_speech(F * 512, sample_offset=0)
case("dense_512_order12_config2", dense, 48000.0, 512, 512, 12, F, dense=True)
case("dense_512_order16", dense, 48000.0, 512, 512, 16, F, dense=True)
timing()
