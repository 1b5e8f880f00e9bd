#!/usr/bin/env python3
"""Where the fused call's time goes at a padded frame length: the spectral kernel ("analyze", HIP-event time of the profile) with and
without MFCC and with / without the formant chain, interpolated bins (default) against the chirp-z kernel beside it (VBX_MFCC_INTERP=0).
usage: python3 tools/experiments/interp_parts.py [--hours 2] [--shapes 1199:480:48000,1103:441:44100]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

def main():
    hours = float(sys.argv[sys.argv.index("--hours") + 1]) if "--hours" in sys.argv else 2.0
    shapes = sys.argv[sys.argv.index("--shapes") + 1] if "--shapes" in sys.argv else "1200:480:48000,1199:480:48000,1103:441:44100,1600:640:48000,3000:1200:48000"
    pkg = g.load_package()
    for sh in shapes.split(","):
        n, hop, sr = sh.split(":"); n, hop, sr = int(n), int(hop), float(sr)
        for mode in ("1", "0"):
            os.environ["VBX_MFCC_INTERP"] = mode
            vb = pkg.VoxBox(0)
            ns = int(hours * 3600 * sr)
            audio = vb.synth_speech(ns)
            F = pkg.frame_count(ns, n, hop)
            est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
            for label, kw in (("pitch+lpc+mfcc+formants", dict(lpc_order=12, formant_order=12, est_init=est0, mfcc=(13, 100.0, 8000.0))),
                              ("pitch+lpc+mfcc", dict(lpc_order=12, formant_order=0, mfcc=(13, 100.0, 8000.0))),
                              ("pitch+lpc", dict(lpc_order=12, formant_order=0, mfcc=None))):
                if mode == "0" and label == "pitch+lpc":
                    continue
                params = pkg.AnalysisParams.make(sr, pitch=(0.2, 75.0, 600.0), **kw)
                REC = (int(vb.L.vbx_record_doubles(params)) + 1) & ~1
                rec = vb.empty((F, REC)); st3 = vb.empty((3, F), np.int32)
                vb.analyze_frames(audio, params, frame_len=n, stride=hop, n_frames=F, out=rec, record_ld=REC, status=st3)
                vb.profile_reset(); vb.profile(True)
                best = 1e30
                for _ in range(3):
                    vb.timer_begin()
                    vb.analyze_frames(audio, params, frame_len=n, stride=hop, n_frames=F, out=rec, record_ld=REC, status=st3)
                    best = min(best, vb.timer_end())
                rep = vb.profile_report(); vb.profile(False)
                top = sorted(((v[0] / max(v[1], 1), k) for k, v in rep.items()), reverse=True)[:4]
                print("%5d/%-4d interp=%s %-24s call %7.2f ms %6.2f M/s | %s" % (n, hop, mode, label, best, F / best / 1e3,
                      "  ".join("%s %.2f" % (k, ms) for ms, k in top)), flush=True)
                rec.free(); st3.free()
            vb.close()

main()
