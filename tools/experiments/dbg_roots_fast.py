"""The conjugate-pair resonance kernel (k_roots_fast.hip) against the reference's iteration (VBX_ROOTS_DIRECT=1):
counts / statuses equal?, worst relative deviation of frequency and bandwidth, frames handed on, time of both.
usage: python tools/experiments/dbg_roots_fast.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g

pkg = g.load_package()
vb = pkg.VoxBox()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
for N, H in ((512, 512), (1200, 480)):
    audio = vb.synth_speech((F - 1) * H + N, sample_offset=3 * 48000)
    seg = np.arange(0, F, 1000, dtype=np.int64)
    res = {}
    for mode in ("0", "1"):
        os.environ["VBX_ROOTS_DIRECT"] = mode
        for rep in range(3):
            vb.sync(); t0 = time.perf_counter()
            r = vb.find_formants(audio, 48000.0, 12, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F)
            vb.sync(); dt = time.perf_counter() - t0
        res[mode] = (r, dt, vb.last_roots_direct_count())
    a, b = res["0"][0], res["1"][0]
    same = np.array_equal(a["status"], b["status"]) and np.array_equal(a["count"], b["count"])
    nz = b["res"] != 0
    dev = np.abs(a["res"] - b["res"])[nz] / np.abs(b["res"])[nz]
    zeros_ok = np.all(a["res"][~nz] == 0)
    trk = np.max(np.abs(a["formants"] - b["formants"]) / np.abs(b["formants"]))
    print(f"N={N}: status/count equal {same}; zero padding equal {zeros_ok}; worst rel deviation res {dev.max():.2e}, tracks {trk:.2e}; "
          f"handed on {res['0'][2]} ({100.0 * res['0'][2] / F:.3f} %); wall incl. D2H {res['0'][1] * 1e3:.1f} vs {res['1'][1] * 1e3:.1f} ms")
    audio.free()
