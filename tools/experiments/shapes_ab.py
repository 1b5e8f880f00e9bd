#!/usr/bin/env python3
"""The fused call (pitch + LPC + MFCC, no formants; and with them) at several frame shapes, for every library build given: best of three, M frames/s.
usage: python3 tools/experiments/shapes_ab.py lib/a.so lib/b.so [--hours 2] [--shapes 512:256,1024:512,2048:1024,4096:2048]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
SR = 48000.0; ns = int(%(hours)f * 3600 * SR); audio = vb.synth_speech(ns)
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES]); res = {}
for n, hop in %(shapes)r:
    F = pkg.frame_count(ns, n, hop)
    for label, kw in (("nf", dict(lpc_order=12, formant_order=0, mfcc=(13, 100.0, 8000.0))), ("all", dict(lpc_order=12, formant_order=12, est_init=est0, mfcc=(13, 100.0, 8000.0)))):
        params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), **kw)
        REC = (int(vb.L.vbx_record_doubles(params)) + 1) & ~1
        rec = vb.empty((F, REC)); st3 = vb.empty((3, F), np.int32); best = 1e30
        for _ in range(4):
            vb.timer_begin(); vb.analyze_frames(audio, params, frame_len=n, stride=hop, n_frames=F, out=rec, record_ld=REC, status=st3); best = min(best, vb.timer_end())
        res["%%d/%%d %%s" %% (n, hop, label)] = [best, F / best / 1e3]
        rec.free(); st3.free()
print("SHAPES_RESULT " + json.dumps(res))
'''
def main():
    args = sys.argv[1:]; hours = 2.0; shapes = "512:256,1024:512,2048:1024,4096:2048,800:320"
    if "--hours" in args: i = args.index("--hours"); hours = float(args[i + 1]); del args[i:i + 2]
    if "--shapes" in args: i = args.index("--shapes"); shapes = args[i + 1]; del args[i:i + 2]
    sh = [tuple(int(v) for v in s.split(":")) for s in shapes.split(",")]
    for lib in args:
        env = dict(os.environ, VBX_LIB_PATH=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "hours": hours, "shapes": sh}], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("SHAPES_RESULT ")]
        if not line: print(lib, "FAILED", p.stderr[-1500:]); continue
        r = json.loads(line[0][14:])
        print("%-28s %s" % (os.path.basename(lib), "  ".join("%s %.2f ms %.2f M" % (k, v[0], v[1]) for k, v in r.items())), flush=True)
main()
