#!/usr/bin/env python3
"""MFCC (13 coefficients, 100..8000 Hz) frames/s against the frame length, hop = frame_len / 2 unless given, on the bench's synthetic speech:
usage: python3 tools/experiments/mfcc_by_len.py [N[:hop] ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    pkg = g.load_package()
    vb = pkg.VoxBox(0)
    specs = sys.argv[1:] or ["400:160", "512:256", "800:320", "1024:512", "1102:441", "1200:480", "1600:640", "2048:1024", "4096:2048"]
    SR = 48000.0
    ns = int(0.5 * 3600 * SR)
    audio = vb.synth_speech(ns)
    rows = []
    for spec in specs:
        n, _, h = spec.partition(":")
        N = int(n); H = int(h) if h else N // 2
        F = min(pkg.frame_count(ns, N, H), 400_000)
        han = vb.window(pkg.WINDOW_HANNING, N)
        out = (vb.empty((F, 13)), vb.empty(F, np.int32))
        best = 1e30
        for _ in range(3):
            vb.timer_begin()
            vb.mfcc(audio, 13, (100.0, 8000.0), SR, frame_len=N, stride=H, n_frames=F, window=han, out=out)
            best = min(best, vb.timer_end())
        rows.append({"frame_len": N, "hop": H, "frames": F, "ms": round(best, 3), "Mframes_per_s": round(F / best / 1e3, 2),
                     "GBps": round(F * H * 8 / best / 1e6, 1)})
        print(json.dumps(rows[-1]), flush=True)
        for d in out:
            d.free()


if __name__ == "__main__":
    main()
