#!/usr/bin/env python3
"""How much of the pitch kernel is the refinement, by frame shape: vbx_pitch_f64 (kmax = 1) with the speech band against the same call with
a band no candidate can pass (the transforms, the peak scan and the filter remain).  usage: python3 tools/experiments/refine_share.py [--hours 2]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
hours = float(sys.argv[sys.argv.index("--hours") + 1]) if "--hours" in sys.argv else 2.0
pkg = g.load_package(); vb = pkg.VoxBox(0)
SR = 48000.0; ns = int(hours * 3600 * SR); audio = vb.synth_speech(ns)
for n, hop in [(1200, 480), (1024, 512), (2048, 1024), (4096, 2048), (4096, 1024), (3000, 1200)]:
    F = pkg.frame_count(ns, n, hop); han = vb.window(pkg.WINDOW_HANNING, n)
    out = (vb.empty((F, 1, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32))
    res = []
    for lo, hi in ((75.0, 600.0), (1e9, 2e9)):
        best = 1e30
        for _ in range(3):
            vb.timer_begin(); vb.pitch(audio, SR, 0.2, lo, hi, kmax=1, frame_len=n, stride=hop, n_frames=F, window=han, out=out); best = min(best, vb.timer_end())
        res.append(best)
    print("%5d/%-5d %8d frames  pitch %7.2f ms (%6.2f ns/frame)   no candidate passes %7.2f ms (%6.2f ns/frame)   refinement share %.2f" % (
        n, hop, F, res[0], res[0] / F * 1e6, res[1], res[1] / F * 1e6, 1 - res[1] / res[0]), flush=True)
    for o in out: o.free()
