#!/usr/bin/env python3
"""Wall time of ONE vbx_analyze_frames_f64 call (everything on: pitch + LPC + formants + MFCC, 1200 / 480 at 48 kHz) against the
batch size -- the serving view of the path: a caller that holds a few hundred frames and wants them back (launch-bound below a
few thousand frames: ~20 kernels per call on three streams).  Median of 50 calls, context warm, input resident.
usage: python3 tools/experiments/latency_by_batch.py [frames ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
vb = pkg.VoxBox(0)
N, H, SR, P = 1200, 480, 48000.0, 12
sizes = [int(a) for a in sys.argv[1:]] or [1, 10, 100, 1000, 10000, 100000, 1000000]
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0, mfcc=(13, 100.0, 8000.0))
REC = int(vb.L.vbx_record_doubles(params))
audio = vb.synth_speech((max(sizes) - 1) * H + N)
for F in sizes:
    rec, st3 = vb.empty((F, REC)), vb.empty((3, F), np.int32)
    call = lambda: (vb.analyze_frames(audio, params, frame_len=N, stride=H, n_frames=F, out=rec, record_ld=REC, status=st3), vb.sync())
    for _ in range(5):
        call()
    ts = []
    for _ in range(50 if F <= 100000 else 10):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    print("frames %8d  median %9.1f us  min %9.1f us  %8.3f M frames/s" % (F, med * 1e6, min(ts) * 1e6, F / med / 1e6), flush=True)
    rec.free(); st3.free()
vb.close()
