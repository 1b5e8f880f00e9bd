#!/usr/bin/env python3
"""vbx_mfcc_f64 alone at frame lengths that do not divide a transform: one forward transform + interpolated bins (default) against the
kernels of rounds 1-4 (VBX_MFCC_INTERP=0: chirp-z / matrix-core DFT / Goertzel): largest difference and the times.
usage: python3 tools/experiments/mfcc_alone_check.py [--hours 1]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

def main():
    hours = float(sys.argv[sys.argv.index("--hours") + 1]) if "--hours" in sys.argv else 1.0
    pkg = g.load_package(); out = {}
    for (n, hop, sr) in [(1103, 441, 44100.0), (1102, 441, 44100.0), (1000, 500, 48000.0), (882, 441, 44100.0), (700, 350, 48000.0), (600, 300, 48000.0), (1280, 640, 48000.0),
                         (1500, 750, 48000.0), (1600, 640, 48000.0), (1800, 900, 48000.0), (2047, 1024, 48000.0), (2500, 1000, 48000.0), (3000, 1200, 48000.0), (4000, 2000, 48000.0)]:
        res = {}
        for mode in ("0", "1"):
            os.environ["VBX_MFCC_INTERP"] = mode
            vb = pkg.VoxBox(0)
            ns = int(hours * 3600 * sr); audio = vb.synth_speech(ns); F = pkg.frame_count(ns, n, hop)
            han = vb.window(pkg.WINDOW_HANNING, n)
            o = vb.empty((F, 13)); st = vb.empty(F, np.int32)
            best = 1e30
            for _ in range(3):
                vb.timer_begin(); vb.mfcc(audio, 13, (100.0, 8000.0), sr, frame_len=n, stride=hop, n_frames=F, window=han, out=(o, st)); best = min(best, vb.timer_end())
            res[mode] = (o.numpy().copy(), st.numpy().copy(), best)
            vb.close()
        d = float(np.abs(res["0"][0] - res["1"][0]).max())
        out["%d/%d" % (n, hop)] = {"frames": F, "max_abs_diff": d, "status_same": bool(np.array_equal(res["0"][1], res["1"][1])), "ms_before": res["0"][2], "ms_interp": res["1"][2],
                                   "Mfps_before": F / res["0"][2] / 1e3, "Mfps_interp": F / res["1"][2] / 1e3}
        print(n, hop, json.dumps(out["%d/%d" % (n, hop)]), flush=True)
main()
