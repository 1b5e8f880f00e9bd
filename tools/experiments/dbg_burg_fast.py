"""One-pass Burg (k_burg_fast.hip) against the direct recursion (VBX_BURG_DIRECT=1) and the oracle: worst deviation in the
parity metric, how many frames the guard sent to the direct kernel, and the time of both forms.
usage: python tools/experiments/dbg_burg_fast.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g

pkg = g.load_package()
oracle = g.load_oracle()
vb = pkg.VoxBox()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
P = 12


def metric(got, exp):
    sc = np.max(np.abs(exp), axis=1, keepdims=True)
    return np.max(np.abs(got - exp) / np.maximum(np.abs(exp), 1e-6 * sc + 1e-300), axis=1)


for N, H in ((512, 512), (1200, 480), (1024, 256), (600, 200), (130, 64), (2048, 1024)):
    audio = vb.synth_speech((F - 1) * H + N, sample_offset=3 * 48000)
    w = vb.window(pkg.WINDOW_HANNING_PERIODIC, N)
    out = (vb.empty((F, P)), vb.empty(F, np.int32))
    res = {}
    for mode in ("0", "1"):
        os.environ["VBX_BURG_DIRECT"] = mode
        for rep in range(3):
            vb.sync()
            t0 = time.perf_counter()
            vb.lpc_praat(audio, P, frame_len=N, stride=H, n_frames=F, window=w, out=out)
            vb.sync()
            dt = time.perf_counter() - t0
        res[mode] = (out[0].numpy(), out[1].numpy(), dt, vb.last_burg_direct_count())
    fast, direct = res["0"], res["1"]
    m = metric(fast[0], direct[0])
    print(f"N={N} H={H} F={F}: status equal {np.array_equal(fast[1], direct[1])}; worst metric vs direct {m.max():.2e} (99.99% {np.quantile(m, 0.9999):.2e}); "
          f"sent to direct {fast[3]} ({100.0 * fast[3] / F:.2f} %); {fast[2] * 1e3:.2f} ms vs direct {direct[2] * 1e3:.2f} ms")
    audio.free()
