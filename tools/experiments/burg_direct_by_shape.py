#!/usr/bin/env python3
"""Round 6: the share of frames the one-pass Burg's guard hands to the direct recursion, per frame shape, on the
synthetic signal -- and what the list's kernel costs beside the fused call.  usage: python3 tools/experiments/burg_direct_by_shape.py [hours]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
SR = 48000.0
SHAPES = ((512, 256), (1024, 512), (1200, 480), (1600, 640), (2048, 1024), (2500, 1000), (3000, 1200), (4000, 2000), (4096, 2048), (4096, 1024))
with pkg.VoxBox(0) as vb:
    ns = int(hours * 3600 * SR)
    audio = vb.synth_speech(ns)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    for n, hop in SHAPES:
        F = pkg.frame_count(ns, n, hop)
        for order in (12,):
            vb.profile_reset(); vb.profile(True)
            ff = vb.find_formants(audio, SR, order, est0, frame_len=n, stride=hop, n_frames=F)
            vb.sync()
            prof = dict(vb.profile_report()); vb.profile(False)
            nb = vb.last_burg_direct_count()
            print(json.dumps({"frame_len": n, "hop": hop, "order": order, "frames": F, "burg_direct": nb, "share": nb / F,
                              "kernels_ms": {k: round(ms / max(c, 1), 3) for k, (ms, c) in prof.items()}}), flush=True)
