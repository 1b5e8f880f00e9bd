#!/usr/bin/env python3
"""What each part of the fused frame loop costs: vbx_analyze_frames_f64 on the bench's shard with parts switched off.
usage: python3 tools/experiments/analyze_parts.py [hours=4]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
vb = pkg.VoxBox(0)
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
SR, N, H = 48000.0, 1200, 480
F = int(hours * 3600 * 100) // 1000 * 1000
audio = vb.synth_speech((F - 1) * H + N)
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
seg = np.arange(0, F, 1000, dtype=np.int64)
for name, kw in (("pitch only", dict(lpc_order=0, formant_order=0, mfcc=None)),
                 ("pitch + LPC", dict(lpc_order=12, formant_order=0, mfcc=None)),
                 ("pitch + MFCC", dict(lpc_order=0, formant_order=0, mfcc=(13, 100.0, 8000.0))),
                 ("pitch + LPC + MFCC", dict(lpc_order=12, formant_order=0, mfcc=(13, 100.0, 8000.0))),
                 ("pitch + formants", dict(lpc_order=0, formant_order=12, mfcc=None)),
                 ("everything", dict(lpc_order=12, formant_order=12, mfcc=(13, 100.0, 8000.0)))):
    params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), est_init=est0, **kw)
    REC = (int(vb.L.vbx_record_doubles(params)) + 1) & ~1
    rec = vb.empty((F, REC)); st3 = vb.empty((3, F), np.int32)
    best = 1e30
    for _ in range(3):
        vb.timer_begin()
        vb.analyze_frames(audio, params, seg_start=seg, frame_len=N, stride=H, n_frames=F, out=rec, record_ld=REC, status=st3)
        best = min(best, vb.timer_end())
    print("%-22s %8.2f ms  %6.2f M frames/s" % (name, best, F / best / 1e3), flush=True)
    rec.free(); st3.free()
