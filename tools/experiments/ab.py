#!/usr/bin/env python3
"""A/B timing of library builds on one GPU box: for every .so given, a child process loads it (VBX_LIB_PATH) and
times the pitch kernel (config-3 shape), the same with no candidates (FFT + peak scan only), the fused analyze call
(pipeline) and checks the top candidates against the first library's (bit-equal expected unless the variant changes
arithmetic).  usage: python3 tools/experiments/ab.py lib/a.so lib/b.so ...   [--hours 0.5]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
N, H, SR = 1200, 480, 48000.0
ns = int(%(hours)f * 3600 * 48000)
audio = vb.synth_speech(ns); F = pkg.frame_count(ns, N, H)
han = vb.window(pkg.WINDOW_HANNING, N)
out = (vb.empty((F, 1, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32))
res = {"frames": F}
def t(label, fn, reps=3):
    best = 1e30
    for i in range(reps):
        vb.timer_begin(); fn(); best = min(best, vb.timer_end())
    res[label] = best
t("pitch_ms", lambda: vb.pitch(audio, SR, 0.2, 75., 600., kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=out))
top = out[0].numpy().copy(); cnt = out[1].numpy().copy()
t("fft_only_ms", lambda: vb.pitch(audio, SR, 0.2, 1e9, 2e9, kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=out))
t("pitch_k64_ms", lambda: vb.pitch(audio, SR, 0.2, 75., 600., kmax=64, frame_len=N, stride=H, n_frames=F // 8, window=han,
                                   out=(vb.empty((F // 8, 64, 2)), vb.empty(F // 8, np.int32), vb.empty(F // 8, np.int32))), reps=2)
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=12, formant_order=12, est_init=est0, mfcc=(13, 100.0, 8000.0))
REC = int(vb.L.vbx_record_doubles(params))
rec = vb.empty((F, REC)); st3 = vb.empty((3, F), np.int32)
seg = np.arange(0, F, 1000, dtype=np.int64)
t("analyze_ms", lambda: vb.analyze_frames(audio, params, seg_start=seg, frame_len=N, stride=H, n_frames=F, out=rec, record_ld=REC, status=st3))
p2 = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=12, formant_order=0, mfcc=(13, 100.0, 8000.0))
REC2 = int(vb.L.vbx_record_doubles(p2))
rec2 = vb.empty((F, REC2))
t("analyze_noformants_ms", lambda: vb.analyze_frames(audio, p2, frame_len=N, stride=H, n_frames=F, out=rec2, record_ld=REC2, status=st3))
np.save(%(save)r, np.concatenate([top.reshape(F, 2), cnt.reshape(F, 1).astype(np.float64)], axis=1))
print("AB_RESULT " + json.dumps(res))
'''

def main():
    args = sys.argv[1:]
    hours = 0.5
    if "--hours" in args:
        i = args.index("--hours"); hours = float(args[i + 1]); del args[i:i + 2]
    ref = None
    for k, spec in enumerate(args):
        # "path/to/lib.so" or "path/to/lib.so:ENV=VAL,ENV2=VAL" (environment of that variant)
        lib, _, envs = spec.partition(":")
        save = "/tmp/ab_%d.npy" % k
        env = dict(os.environ, VBX_LIB_PATH=os.path.abspath(lib))
        for kv in filter(None, envs.split(",")):
            a, _, b = kv.partition("="); env[a] = b
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "hours": hours, "save": save}], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("AB_RESULT ")]
        if not line:
            print(spec, "FAILED", p.stdout[-2000:], p.stderr[-2000:]); continue
        r = json.loads(line[0][10:])
        import numpy as np
        cur = np.load(save)
        if ref is None:
            ref = cur; same = "reference"
        else:
            nd = int(np.sum(np.any(cur != ref, axis=1)))
            bad = int(np.sum((np.abs(cur[:, 0] - ref[:, 0]) > 1e-4 * np.abs(ref[:, 0])) | (np.abs(cur[:, 1] - ref[:, 1]) > 1e-4) | (cur[:, 2] != ref[:, 2])))
            same = f"{nd} frames differ bitwise, {bad} beyond 1e-4 / count"
        F = r["frames"]
        print("%-44s pitch %8.3f ms (%6.2f M/s)  fft-only %7.3f  k64(F/8) %8.3f  analyze %8.3f ms (%6.2f M/s)  no-formants %7.3f  | %s" % (
            os.path.basename(lib) + (":" + envs if envs else ""), r["pitch_ms"], F / r["pitch_ms"] / 1e3, r["fft_only_ms"], r["pitch_k64_ms"], r["analyze_ms"],
            F / r["analyze_ms"] / 1e3, r.get("analyze_noformants_ms", 0.0), same), flush=True)

main()
