#!/usr/bin/env python3
"""Context create / use / destroy in a loop: device memory in use must not grow (workspaces, tables, streams, events)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
import ctypes as C

pkg = g.load_package()
hip = C.CDLL("libamdhip64.so")


def free_bytes():
    a, b = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(a), C.byref(b)) == 0
    return a.value


SR = 48000.0
rng = np.random.default_rng(1)
first = None
for it in range(30):
    vb = pkg.VoxBox(0)
    for N, H in ((1200, 480), (1024, 512), (2048, 1024), (4096, 2048), (700, 300), (400, 160)):
        F = 3000
        audio = vb.synth_speech((F - 1) * H + N)
        han = vb.window(pkg.WINDOW_HANNING, N)
        vb.pitch(audio, SR, 0.2, 75.0, 600.0, kmax=4, frame_len=N, stride=H, n_frames=F, window=han)
        vb.mfcc(audio, 13, (100.0, 8000.0), SR, frame_len=N, stride=H, n_frames=F, window=han)
        vb.autocorrelate(audio, min(N, 300), frame_len=N, stride=H, n_frames=F, window=han)
        est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
        vb.find_formants(audio, SR, 12, est0, seg_start=np.arange(0, F, 500, dtype=np.int64), frame_len=N, stride=H, n_frames=F)
        params = pkg.AnalysisParams.make(SR)
        vb.analyze_frames(audio, params, frame_len=N, stride=H, n_frames=F)
        audio.free()
    vb.close()
    fb = free_bytes()
    if it == 2:
        first = fb
    if it in (2, 10, 20, 29):
        print("iteration", it, "free MB", fb >> 20, flush=True)
print("growth after warm-up (MB):", (first - fb) >> 20)
assert first - fb < (64 << 20), "device memory in use grows"
print("leak check ok")
