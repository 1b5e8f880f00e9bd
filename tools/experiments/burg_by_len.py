#!/usr/bin/env python3
"""LPC::lpc_praat (Burg, order 12) frames/s and GB/s against the frame length (dense frames: hop = N).
usage: python3 tools/experiments/burg_by_len.py [N ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
vb = pkg.VoxBox(0)
Ns = [int(a) for a in sys.argv[1:]] or [128, 256, 400, 512, 800, 1024, 1200, 1600, 2048, 4096]
ns = 200_000_000
audio = vb.synth_speech(ns)
for N in Ns:
    F = min(ns // N, 400_000)
    han = vb.window(pkg.WINDOW_HANNING_PERIODIC, N)
    out = (vb.empty((F, 12)), vb.empty(F, np.int32))
    best = 1e30
    for _ in range(3):
        vb.timer_begin()
        vb.lpc_praat(audio, 12, frame_len=N, stride=N, n_frames=F, window=han, out=out)
        best = min(best, vb.timer_end())
    print(json.dumps({"frame_len": N, "frames": F, "ms": round(best, 3), "Mframes_per_s": round(F / best / 1e3, 2),
                      "GBps": round(F * N * 8 / best / 1e6, 1), "Msamples_per_s": round(F * N / best / 1e3, 1)}), flush=True)
    for d in out:
        d.free()
