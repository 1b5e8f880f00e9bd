#!/usr/bin/env python3
"""The 4096-point plan as two kernels (transforms + LPC + MFCC, then the refinement from a scratch row: SP_ANALYZE_SPLIT) against the
fused kernel (VBX_POW2_SPLIT=0): every column of the fused call's records, vbx_pitch_f64 at kmax 1 and 8, statuses and counts, bit for
bit; and the times.  usage: python3 tools/experiments/split_check.py [--hours 2]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

def main():
    hours = float(sys.argv[sys.argv.index("--hours") + 1]) if "--hours" in sys.argv else 2.0
    pkg = g.load_package(); out = {}
    SR = 48000.0
    shapes = [(4096, 2048), (4096, 1024), (3000, 1200), (4000, 2000), (2500, 1000), (2050, 1024), (4095, 2048)]
    if "--shapes" in sys.argv:
        shapes = [tuple(int(v) for v in s.split(":")) for s in sys.argv[sys.argv.index("--shapes") + 1].split(",")]
    for (n, hop) in shapes:
        res = {}
        for mode in ("0", "1"):
            os.environ["VBX_POW2_SPLIT"] = mode
            vb = pkg.VoxBox(0)
            ns = int(hours * 3600 * SR); audio = vb.synth_speech(ns); F = pkg.frame_count(ns, n, hop)
            est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
            params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=12, formant_order=12, est_init=est0, mfcc=(13, 100.0, 8000.0))
            REC = (int(vb.L.vbx_record_doubles(params)) + 1) & ~1
            rec = vb.empty((F, REC)); st3 = vb.empty((3, F), np.int32)
            han = vb.window(pkg.WINDOW_HANNING, n)
            o1 = (vb.empty((F, 1, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32))
            F8 = min(F, 40000)
            o8 = (vb.empty((F8, 8, 2)), vb.empty(F8, np.int32), vb.empty(F8, np.int32))
            t = {}
            for label, fn in (("analyze", lambda: vb.analyze_frames(audio, params, frame_len=n, stride=hop, n_frames=F, out=rec, record_ld=REC, status=st3)),
                              ("pitch", lambda: vb.pitch(audio, SR, 0.2, 75., 600., kmax=1, frame_len=n, stride=hop, n_frames=F, window=han, out=o1)),
                              ("pitch_k8", lambda: vb.pitch(audio, SR, 0.2, 75., 600., kmax=8, frame_len=n, stride=hop, n_frames=F8, window=han, out=o8))):
                best = 1e30
                for _ in range(3):
                    vb.timer_begin(); fn(); best = min(best, vb.timer_end())
                t[label] = best
            res[mode] = ([rec.numpy().copy(), st3.numpy().copy()] + [o.numpy().copy() for o in o1] + [o.numpy().copy() for o in o8], t, F)
            vb.close()
        same = all(np.array_equal(a, b) for a, b in zip(res["0"][0], res["1"][0]))
        F = res["0"][2]
        out["%d/%d" % (n, hop)] = {"frames": F, "bit_identical": bool(same), "fused_ms": res["0"][1], "split_ms": res["1"][1],
                                   "Mfps_fused": F / res["0"][1]["analyze"] / 1e3, "Mfps_split": F / res["1"][1]["analyze"] / 1e3}
        print(n, hop, json.dumps(out["%d/%d" % (n, hop)]), flush=True)
    print("SPLIT_REPORT " + json.dumps(out))
main()
