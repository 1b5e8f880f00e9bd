#!/usr/bin/env python3
"""The formant chain on its own (vbx_find_formants_f64: Burg -> roots -> tracker), per-kernel HIP-event times, at several frame shapes:
what the fused call has to hide beside its spectral kernel.  usage: python3 tools/experiments/formants_alone.py [--hours 1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
hours = float(sys.argv[sys.argv.index("--hours") + 1]) if "--hours" in sys.argv else 1.0
pkg = g.load_package(); vb = pkg.VoxBox(0)
for n, hop, sr in [(1200, 480, 48000.0), (1199, 480, 48000.0), (1103, 441, 44100.0), (1102, 441, 44100.0), (1104, 440, 44100.0), (1024, 512, 48000.0), (800, 320, 48000.0), (1600, 640, 48000.0)]:
    ns = int(hours * 3600 * sr)
    audio = vb.synth_speech(ns); F = pkg.frame_count(ns, n, hop)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    vb.find_formants(audio, sr, 12, est0, frame_len=n, stride=hop, n_frames=F)
    vb.profile_reset(); vb.profile(True)
    vb.timer_begin(); vb.find_formants(audio, sr, 12, est0, frame_len=n, stride=hop, n_frames=F); ms = vb.timer_end()
    rep = vb.profile_report(); vb.profile(False)
    print("%5d/%-4d %7d frames  call %6.2f ms | %s" % (n, hop, F, ms, "  ".join("%s %.2f" % (k, v[0]) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])[:6])), flush=True)
    audio.free()
