#!/usr/bin/env python3
"""Where a wavefront of the fused kernel spends its life (experiment build -DVBX_EXP_PHASES, lib/libvoxbox_hip_phases.so):
shader-clock cycles per phase, per frame.  usage: VBX_LIB_PATH=.../libvoxbox_hip_phases.so python3 tools/experiments/phases.py [hours]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
N, H, SR = (int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (1200, 480, 48000.0)
ns = int(hours * 3600 * SR)
audio = vb.synth_speech(ns); F = pkg.frame_count(ns, N, H)
han = vb.window(pkg.WINDOW_HANNING, N)
out = (vb.empty((F, 1, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32))
def run(label, fn):
    vb.profile_reset(); vb.profile(True)
    vb.timer_begin(); fn(); ms = vb.timer_end()
    sys.stderr.write("VBX_PHASES_LABEL %s %d frames %.3f ms\n" % (label, F, ms)); sys.stderr.flush()
    vb.profile_pitch_work()
    vb.profile(False)
run("pitch", lambda: vb.pitch(audio, SR, 0.2, 75., 600., kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=out))
p2 = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=12, formant_order=0, mfcc=(13, 100.0, 8000.0))
REC2 = (int(vb.L.vbx_record_doubles(p2)) + 1) & ~1
rec2 = vb.empty((F, REC2)); st3 = vb.empty((3, F), np.int32)
run("fused_noformants", lambda: vb.analyze_frames(audio, p2, frame_len=N, stride=H, n_frames=F, out=rec2, record_ld=REC2, status=st3))
