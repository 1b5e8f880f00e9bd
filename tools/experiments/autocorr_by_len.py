#!/usr/bin/env python3
"""Autocorrelate::autocorrelate(n_lags) frames/s for many lags (the few-lag kernel serves n_lags <= 17).
usage: python3 tools/experiments/autocorr_by_len.py [N:lags ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
vb = pkg.VoxBox(0)
specs = sys.argv[1:] or ["512:64", "512:512", "1200:100", "1200:1200", "2048:256", "2048:2048", "4096:4096"]
ns = int(0.25 * 3600 * 48000)
audio = vb.synth_speech(ns)
for spec in specs:
    n, _, l = spec.partition(":")
    N, L = int(n), int(l)
    H = N // 2
    F = min(pkg.frame_count(ns, N, H), 100_000)
    han = vb.window(pkg.WINDOW_HANNING, N)
    out = vb.empty((F, L))
    best = 1e30
    for _ in range(3):
        vb.timer_begin()
        vb.autocorrelate(audio, L, frame_len=N, stride=H, n_frames=F, window=han, out=out)
        best = min(best, vb.timer_end())
    print(json.dumps({"frame_len": N, "lags": L, "frames": F, "ms": round(best, 3), "Mframes_per_s": round(F / best / 1e3, 2)}), flush=True)
    out.free()
