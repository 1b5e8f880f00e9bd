#!/usr/bin/env python3
"""How many frames the FFT pitch kernels hand to the direct lag-sum kernel (peak decisions inside the transform's rounding
error), by frame shape, on the bench signal.  usage: python3 tools/experiments/unsure_by_len.py [N:hop ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
SR = 48000.0
ns = int(0.25 * 3600 * SR)
audio = vb.synth_speech(ns)
for spec in (sys.argv[1:] or ["1200:480", "1600:640", "2048:1024", "2500:1000", "3000:1200", "3500:1400", "4096:2048"]):
    N, H = (int(v) for v in spec.split(":"))
    F = pkg.frame_count(ns, N, H)
    han = vb.window(pkg.WINDOW_HANNING, N)
    out = (vb.empty((F, 1, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32))
    vb.pitch(audio, SR, 0.2, 75., 600., kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=out)
    vb.sync()
    print(spec, "frames", F, "deferred to the direct kernel:", vb.last_unsure_count(), flush=True)
    for d in out: d.free()
