#!/usr/bin/env python3
"""Driver for a counter pass (rocprofv3 --pmc ... -- python3 tools/experiments/valu_parts.py): the pitch kernel of the
headline shape launched in forms that stop earlier and earlier, each ONCE and in this order, so that the per-dispatch
counter rows can be told apart:
  1. pitch, kmax = 1 (everything)            2. no candidate passes the filter (fmin = 1e9: transforms + scan only)
  3. every frame unvoiced-like? -- not separable from outside; see (2)
  4. the fused analyze call without formants 5. MFCC alone 6. autocorrelate(1200) alone (both transforms, no scan)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
N, H, SR = 1200, 480, 48000.0
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
ns = int(hours * 3600 * 48000)
audio = vb.synth_speech(ns); F = pkg.frame_count(ns, N, H)
han = vb.window(pkg.WINDOW_HANNING, N)
out = (vb.empty((F, 1, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32))
vb.pitch(audio, SR, 0.2, 75., 600., kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=out)
vb.sync()
vb.pitch(audio, SR, 0.2, 1e9, 2e9, kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=out)
vb.sync()
p2 = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=12, formant_order=0, mfcc=(13, 100.0, 8000.0))
REC2 = int(vb.L.vbx_record_doubles(p2))
rec2 = vb.empty((F, REC2)); st3 = vb.empty((3, F), np.int32)
vb.analyze_frames(audio, p2, frame_len=N, stride=H, n_frames=F, out=rec2, record_ld=REC2, status=st3)
vb.sync()
mf = vb.empty((F, 13)); ms = vb.empty(F, np.int32)
vb.mfcc(audio, 13, (100.0, 8000.0), SR, frame_len=N, stride=H, n_frames=F, window=han, out=(mf, ms))
vb.sync()
r = vb.empty((F // 4, N))
vb.autocorrelate(audio, N, frame_len=N, stride=H, n_frames=F // 4, window=han, out=r)
vb.sync()
print("frames", F)
