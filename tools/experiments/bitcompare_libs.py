#!/usr/bin/env python3
"""Every column of the fused call's record, of vbx_pitch_f64 (kmax 1 and 8) and of vbx_find_formants_f64 (Burg rows, resonance rows,
tracks, statuses), bit for bit between two builds of the library, on an hour of the bench's recording at several frame shapes.
usage: python3 tools/experiments/bitcompare_libs.py lib/a.so lib/b.so [--hours 1]      (each build runs in a child process)"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(1200, 480), (1024, 512), (2048, 1024), (4096, 2048), (1103, 441), (800, 320), (512, 256)]

CHILD = r'''
import sys, json, hashlib, numpy as np
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
SR = 48000.0
ns = int(%(hours)f * 3600 * 48000)
audio = vb.synth_speech(ns)
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
out = {}
def dig(a): return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
for n, hop in %(shapes)r:
    F = min(pkg.frame_count(ns, n, hop), 200000)
    params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=12, formant_order=12, est_init=est0, mfcc=(13, 100.0, 8000.0))
    rec, st3 = vb.analyze_frames(audio, params, frame_len=n, stride=hop, n_frames=F)
    han = vb.window(pkg.WINDOW_HANNING, n)
    c1, n1, s1 = vb.pitch(audio, SR, 0.2, 75., 600., kmax=1, frame_len=n, stride=hop, n_frames=F, window=han)
    c8, n8, s8 = vb.pitch(audio, SR, 0.2, 75., 600., kmax=8, frame_len=n, stride=hop, n_frames=min(F, 50000), window=han)
    ff = vb.find_formants(audio, SR, 12, est0, frame_len=n, stride=hop, n_frames=F)
    cols = params.columns()
    d = {"frames": F, "status3": dig(st3), "pitch1": dig(c1), "count1": dig(n1), "pitch8": dig(c8)}
    for k, (c0, w) in cols.items(): d["record_" + k] = dig(rec[:, c0:c0 + w])
    for k in ("formants", "res", "count", "coeffs", "status"): d["ff_" + k] = dig(ff[k])
    out["%%d/%%d" %% (n, hop)] = d
print("BITCMP " + json.dumps(out))
'''

def main():
    args = sys.argv[1:]
    hours = 1.0
    if "--hours" in args:
        i = args.index("--hours"); hours = float(args[i + 1]); del args[i:i + 2]
    res = []
    for lib in args:
        env = dict(os.environ, VBX_LIB_PATH=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "hours": hours, "shapes": SHAPES}], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("BITCMP ")]
        if not line:
            print(lib, "FAILED", p.stdout[-1500:], p.stderr[-1500:]); return 1
        res.append(json.loads(line[0][7:]))
    rep = {"libs": [os.path.basename(a) for a in args], "hours": hours, "shapes": {}}
    for shape in res[0]:
        diff = [k for k in res[0][shape] if any(r[shape][k] != res[0][shape][k] for r in res[1:])]
        rep["shapes"][shape] = {"frames": res[0][shape]["frames"], "columns_compared": len(res[0][shape]) - 1, "columns_that_differ": diff}
        print(shape, res[0][shape]["frames"], "frames:", "IDENTICAL" if not diff else "DIFFER in %s" % diff)
    print("BITCMP_REPORT " + json.dumps(rep))
    return 0

sys.exit(main())
