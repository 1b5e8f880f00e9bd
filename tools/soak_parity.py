#!/usr/bin/env python3
"""Parity at scale: the GPU's PitchExtractor output, candidate count and status, Burg coefficients, formant resonances and
MFCC against the CPU oracle on tens of thousands of frames (the pytest suite compares hundreds).  The oracle runs on all
the cores the process may use (ctypes releases the GIL).  Prints and writes a JSON summary of every disagreement class.
The pitch candidates come from vbx_pitch_f64 AND from the fused frame loop (vbx_analyze_frames_f64: pitch + LPC + MFCC from
one FFT of the frame where the frame length has such a kernel), whose LPC and MFCC columns are checked too.
usage (GPU box): python3 tools/soak_parity.py [n_frames=20000] [out.json] [frame_len=1200] [hop=480] [visit_step=7] [first_frame=0]
visit_step = 1 checks EVERY frame of a stretch of the recording (first_frame .. first_frame + n_frames): a whole bench shard can be
walked in pieces (tools/soak_shard.sh)."""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

SR, P = 48000.0, 12


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "soak_parity.json")
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
    H = int(sys.argv[4]) if len(sys.argv) > 4 else 480
    step = int(sys.argv[5]) if len(sys.argv) > 5 else 7
    first = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    pkg, o = g.load_package(), g.load_oracle()
    vb = pkg.VoxBox(0)
    try:
        workers = len(os.sched_getaffinity(0))
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            workers = max(1, min(workers, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError, AttributeError):
        workers = os.cpu_count() or 1
    # frames: the bench's synthetic recording, visited with a stride coprime to the 5 s voiced/unvoiced pattern
    ns = (n_frames * step - 1) * H + N
    audio_d = vb.synth_speech(ns, sample_offset=5 * 48000 + first * H)
    han = vb.window(pkg.WINDOW_HANNING, N)
    F_all = pkg.frame_count(ns, N, H)
    idx = (np.arange(n_frames) * step) % F_all
    cand, cnt, st = vb.pitch(audio_d, SR, 0.2, 75.0, 600.0, kmax=1, frame_len=N, stride=H, n_frames=F_all, window=han)
    co, cst = vb.lpc_praat(audio_d, P, frame_len=N, stride=H, n_frames=F_all, window=vb.window(pkg.WINDOW_HANNING_PERIODIC, N))
    mf, mst = vb.mfcc(audio_d, 13, (100.0, 8000.0), SR, frame_len=N, stride=H, n_frames=F_all, window=han)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    ff = vb.find_formants(audio_d, SR, P, est0, seg_start=np.arange(0, F_all, 1, dtype=np.int64), frame_len=N, stride=H, n_frames=F_all,
                          want=("res", "count", "status"))
    params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=0, mfcc=(13, 100.0, 8000.0))
    cols = params.columns()
    rec_all, st_all = vb.analyze_frames(audio_d, params, frame_len=N, stride=H, n_frames=F_all)
    wh = o.window("hanning", N)
    # every lag of every 8th visited frame (vbx_autocorrelate_f64 with n_lags = N: the FFT kernels from 512 samples on)
    ac_idx = idx[::8]
    ac_frames = np.stack([audio_d.numpy_slice(int(t) * H, N) for t in ac_idx])
    ac_gpu = vb.autocorrelate(ac_frames * wh, N)
    ac_pos = {int(t): k for k, t in enumerate(ac_idx)}

    def one(t):
        fr = audio_d.numpy_slice(int(t) * H, N)
        es, ec, en = o.pitch(fr * wh, SR, 0.2, 75.0, 600.0)
        rec = {"status": int(es != st[t]), "count": int((en if es == 0 else 0) != cnt[t]), "top_bad": 0, "top_swap": 0, "vuv": 0}
        if es == 0:
            ok = abs(cand[t, 0, 0] - ec[0, 0]) <= 1e-4 * abs(ec[0, 0]) and abs(cand[t, 0, 1] - ec[0, 1]) <= 1e-4
            if not ok:
                gap = abs(ec[0, 1] - ec[1, 1]) if en > 1 else np.inf
                runner = en > 1 and abs(cand[t, 0, 0] - ec[1, 0]) <= 1e-4 * abs(ec[1, 0]) and abs(cand[t, 0, 1] - ec[1, 1]) <= 1e-3
                if gap < 1e-3 and runner:
                    rec["top_swap"] = 1
                    rec["vuv"] = int((cand[t, 0, 0] == 0.0) != (ec[0, 0] == 0.0) and gap > 1e-4)
                else:
                    rec["top_bad"] = 1
        bs, bc = o.lpc_burg(fr * o.window("hanning_periodic", N), P)
        sc = np.max(np.abs(bc)) if bs == 0 else 1.0
        rec["burg"] = int(bs != cst[t] or (bs == 0 and not np.all(np.abs(co[t] - bc) <= 1e-6 * np.maximum(np.abs(bc), 1e-6 * sc))))
        fs, _, eres, _ = o.find_formants(fr, SR, P, est0)
        n_res = int(np.sum(eres[:, 0] != 0.0))
        rec["formant"] = int(fs != ff["status"][t] or n_res != ff["count"][t] or
                             not np.all(np.abs(ff["res"][t, :n_res, 0] - eres[:n_res, 0]) <= 1e-4 * np.abs(eres[:n_res, 0])))
        ms, em = o.mfcc(fr * wh, 13, 100.0, 8000.0, SR)
        rec["mfcc"] = int(ms != mst[t] or not np.all(np.abs(mf[t] - em) <= 1e-6 * np.maximum(np.abs(em), 1e-6 * np.max(np.abs(em)))))
        # the fused frame loop's record of the same frame
        c0, cn = cols["pitch"]; m0, mn = cols["mfcc"]; l0, ln = cols["lpc"]
        fm = rec_all[t, m0:m0 + mn]
        rec["fused_mfcc"] = int(ms != st_all[2, t] or not np.all(np.abs(fm - em) <= 1e-6 * np.maximum(np.abs(em), 1e-6 * np.max(np.abs(em)))))
        el = o.lpc(o.autocorrelate(fr * wh, P + 1), P)
        fl = rec_all[t, l0:l0 + ln]
        rec["fused_lpc"] = int(not np.all(np.abs(fl - el) <= 1e-6 * np.maximum(np.abs(el), 1e-6 * np.max(np.abs(el)))))
        rec["fused_lpc_beyond_oracle_rounding"] = 0
        if rec["fused_lpc"]:
            # a coefficient below 1e-6 of the largest one is held to an ABSOLUTE 1e-12: who is right?  The same recursion in
            # long double on long-double lag sums, and both results' distance from it in the same metric
            xl = (fr * wh).astype(np.longdouble)
            rl = np.array([xl[0] + np.sum(xl[1:N - k] * xl[1 + k:N]) for k in range(P + 1)])
            al = np.zeros(P + 1, dtype=np.longdouble); al[0] = 1; err = rl[0]
            for i in range(1, P + 1):
                kk = -(rl[i] + sum(al[j] * rl[i - j] for j in range(1, i))) / err
                tl = al.copy(); al[i] = kk
                for j in range(1, i):
                    al[j] = tl[j] + kk * tl[i - j]
                err = err * (1 - kk * kk)
            dev = lambda v: float(np.max(np.abs(v - al) / np.maximum(np.abs(al), 1e-6 * np.max(np.abs(al)))))
            rec["fused_lpc_beyond_oracle_rounding"] = int(dev(fl) > max(1e-6, 2.0 * dev(el)))
        rec["autocorr_all_lags"] = 0
        if int(t) in ac_pos:
            ea = o.autocorrelate(fr * wh, N)
            ga = ac_gpu[ac_pos[int(t)]]
            rec["autocorr_all_lags"] = int(not np.all(np.abs(ga - ea) <= 1e-6 * np.maximum(np.abs(ea), 1e-6 * np.max(np.abs(ea)))))
        fp = rec_all[t, c0:c0 + 2]
        rec["fused_pitch"] = int(es != st_all[0, t] or (es == 0 and not rec["top_swap"] and not (
            abs(fp[0] - ec[0, 0]) <= 1e-4 * abs(ec[0, 0]) and abs(fp[1] - ec[0, 1]) <= 1e-4)))
        return rec

    with ThreadPoolExecutor(workers) as ex:
        recs = list(ex.map(one, idx))
    tot = {k: int(sum(r[k] for r in recs)) for k in recs[0]}
    voiced = int(np.sum(cand[idx, 0, 0] > 0))
    summary = {"frame_len": N, "hop": H, "visit_step": step, "first_frame": first, "frames": n_frames, "voiced": voiced, "unvoiced": n_frames - voiced, "oracle_threads": workers,
               "disagreements": tot,
               "meaning": {"status": "pitch status differs", "count": "pitch candidate count differs",
                           "top_bad": "PitchExtractor output beyond 1e-4 and not a near tie",
                           "top_swap": "top two oracle strengths closer than 1e-3 and the GPU's top is the runner-up",
                           "vuv": "of the swaps: voiced/unvoiced flips outside a 1e-4 tie",
                           "burg": "Burg status or coefficients beyond 1e-6", "formant": "resonance count or Hz beyond 1e-4",
                           "mfcc": "MFCC status or values beyond 1e-6",
                           "autocorr_all_lags": "autocorrelate(N) of every 8th frame beyond 1e-6 (floor 1e-6 of the row's largest)",
                           "fused_pitch / fused_lpc / fused_mfcc": "the same checks on the columns of vbx_analyze_frames_f64 "
                                                                   "(LPC: Levinson order 12 on autocorrelate(13), 1e-6)",
                           "fused_lpc_beyond_oracle_rounding": "of the fused_lpc frames (a coefficient below 1e-6 of the largest one is "
                                                               "held to an absolute 1e-12 there): those where the GPU is further from "
                                                               "the long-double answer than 1e-6 AND than twice the oracle's own distance"}}
    print(json.dumps(summary))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(summary, open(out_path, "w"), indent=1)
    vb.close()


if __name__ == "__main__":
    main()
