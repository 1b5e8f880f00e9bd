"""One-pass Burg and conjugate-pair roots at every instantiated order: frames sent to the direct recursion / redone by the
reference's iteration, worst deviation from the direct forms, find_formants rate.  usage: python tools/experiments/burg_orders.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox()
F, N, H = 400000, 512, 512
audio = vb.synth_speech((F - 1) * H + N, sample_offset=3 * 48000)
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
seg = np.arange(0, F, 1000, dtype=np.int64)
for p in (8, 10, 12, 13, 14, 16, 11):
    res = {}
    for mode in ("0", "1"):
        os.environ["VBX_BURG_DIRECT"] = mode; os.environ["VBX_ROOTS_DIRECT"] = mode
        best = 1e9
        for rep in range(3):
            vb.timer_begin()
            r = vb.find_formants(audio, 48000.0, p, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F)
            best = min(best, vb.timer_end())
        res[mode] = (r, best, vb.last_burg_direct_count(), vb.last_roots_direct_count())
    a, b = res["0"][0], res["1"][0]
    nz = b["res"] != 0
    dev = float(np.max(np.abs(a["res"] - b["res"])[nz] / np.abs(b["res"])[nz]))
    sc = np.max(np.abs(b["coeffs"]), axis=1, keepdims=True)
    cm = float(np.max(np.abs(a["coeffs"] - b["coeffs"]) / np.maximum(np.abs(b["coeffs"]), 1e-6 * sc)))
    print(f"order {p}: Burg sent to direct {res['0'][2]} ({100.0 * max(res['0'][2], 0) / F:.2f} %), roots redone {res['0'][3]}; coefficients within {cm:.1e}, "
          f"resonances within {dev:.1e}; status/count equal {np.array_equal(a['status'], b['status']) and np.array_equal(a['count'], b['count'])}; "
          f"call incl. D2H {res['0'][1]:.1f} ms vs {res['1'][1]:.1f} ms", flush=True)
