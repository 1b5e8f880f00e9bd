#!/usr/bin/env python3
"""MFCC::mfcc inside the fused kernel at lengths that do not divide the transform (interpolated bins, mfcc_interp_t) against the
chirp-z kernel beside it (VBX_MFCC_INTERP=0): largest difference of the MFCC columns, every other column bit for bit, and the
time of the fused call.  usage: python3 tools/experiments/mfcc_interp_check.py [--hours 0.5]"""
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

def main():
    hours = float(sys.argv[sys.argv.index("--hours") + 1]) if "--hours" in sys.argv else 0.25
    pkg = g.load_package()
    out = {}
    for (n, hop, sr) in [(1103, 441, 44100.0), (1102, 441, 44100.0), (1103, 441, 48000.0), (1025, 512, 48000.0), (1199, 480, 48000.0), (882, 441, 44100.0), (1000, 500, 48000.0),
                         (700, 350, 48000.0), (1201, 600, 48000.0), (1600, 640, 48000.0), (2047, 1024, 48000.0), (2049, 1024, 48000.0), (3000, 1200, 48000.0), (4000, 2000, 48000.0), (4095, 2048, 48000.0)]:
        res = {}
        for mode in ("0", "1"):
            os.environ["VBX_MFCC_INTERP"] = mode
            vb = pkg.VoxBox(0)
            ns = int(hours * 3600 * sr)
            audio = vb.synth_speech(ns)
            F = pkg.frame_count(ns, n, hop)
            est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
            params = pkg.AnalysisParams.make(sr, pitch=(0.2, 75.0, 600.0), lpc_order=12, formant_order=12, est_init=est0, mfcc=(13, 100.0, 8000.0))
            REC = int(vb.L.vbx_record_doubles(params))
            rec = vb.empty((F, REC)); st3 = vb.empty((3, F), np.int32)
            best = 1e30
            for _ in range(3):
                vb.timer_begin()
                vb.analyze_frames(audio, params, frame_len=n, stride=hop, n_frames=F, out=rec, record_ld=REC, status=st3)
                best = min(best, vb.timer_end())
            res[mode] = (rec.numpy().copy(), st3.numpy().copy(), best, F)
            vb.close()
        a, b = res["0"][0], res["1"][0]
        # columns: pitch 2 | formants 2 n_est | mfcc 13 | lpc 13
        n_est = len(pkg.MALE_FORMANT_ESTIMATES)
        c0 = 2 + 2 * n_est
        mf_a, mf_b = a[:, c0:c0 + 13], b[:, c0:c0 + 13]
        other_same = bool(np.array_equal(a[:, :c0], b[:, :c0]) and np.array_equal(a[:, c0 + 13:], b[:, c0 + 13:]))
        d = np.abs(mf_a - mf_b)
        F = res["0"][3]
        out["%d/%d@%g" % (n, hop, sr)] = {"frames": F, "mfcc_max_abs_diff": float(d.max()), "mfcc_scale": float(np.abs(mf_a).max()),
                                          "other_columns_identical": other_same, "status_identical": bool(np.array_equal(res["0"][1], res["1"][1])),
                                          "ms_czt_beside": res["0"][2], "ms_interp": res["1"][2],
                                          "Mfps_czt_beside": F / res["0"][2] / 1e3, "Mfps_interp": F / res["1"][2] / 1e3}
        print(n, hop, sr, json.dumps(out["%d/%d@%g" % (n, hop, sr)]), flush=True)
    print("INTERP_REPORT " + json.dumps(out))

main()
