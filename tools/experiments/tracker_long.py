#!/usr/bin/env python3
"""One long utterance through the tracker: the sequential scan (one lane, ~5.4 us per frame) against the chunked scan
(speculative chunks + exact repair).  usage: python3 tools/experiments/tracker_long.py [frames=360000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
vb = pkg.VoxBox(0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 360000
SR, N, H, P = 48000.0, 512, 512, 12
audio = vb.synth_speech(F * H + N)
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
bufs = {"formants": vb.empty((F, 4, 2)), "res": vb.empty((F, 32, 2)), "count": vb.empty(F, np.int32), "coeffs": None,
        "status": vb.empty(F, np.int32)}
out = {}
for mode in ("0", "1"):
    os.environ["VBX_TRACKER_CHUNKED"] = mode
    best = 1e30
    for _ in range(2):
        vb.sync(); t0 = time.perf_counter()
        vb.find_formants(audio, SR, P, est0, frame_len=N, stride=H, n_frames=F, out=bufs)
        vb.sync(); best = min(best, time.perf_counter() - t0)
    out[mode] = bufs["formants"].numpy().copy()
    print("chunked" if mode == "1" else "sequential", "find_formants of one %d-frame utterance: %.2f ms" % (F, best * 1e3))
print("bit-identical:", np.array_equal(out["0"].view(np.uint64), out["1"].view(np.uint64)))
vb.profile(True)
vb.find_formants(audio, SR, P, est0, frame_len=N, stride=H, n_frames=F, out=bufs)
vb.sync()
print(dict(vb.profile_report()))
