"""Pitch kernel time per frame on voiced-only, noise-only and the bench mix (which frames cost what).
usage: python tools/experiments/pitch_by_signal.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox()
N, H, SR = 1200, 480, 48000.0
secs = 600
han = vb.window(pkg.WINDOW_HANNING, N)
mix = vb.synth_speech(secs * 48000).numpy()
sec = (np.arange(mix.size) // 48000) % 5
voiced = mix[sec != 4][: (secs * 48000 * 3) // 5]
noise = np.tile(mix[sec == 4], 4)[: voiced.size]
rng = np.random.default_rng(0)
white = 0.3 * rng.standard_normal(voiced.size)
for name, sig in (("bench mix", mix[: voiced.size]), ("voiced seconds only", voiced), ("noise seconds only", noise), ("white noise 0.3", white)):
    d = vb.to_device(np.ascontiguousarray(sig))
    F = pkg.frame_count(sig.size, N, H)
    out = (vb.empty((F, 1, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32))
    best = 1e9
    for _ in range(3):
        vb.timer_begin(); vb.pitch(d, SR, 0.2, 75.0, 600.0, kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=out); best = min(best, vb.timer_end())
    top = out[0].numpy(); cnt = out[1].numpy()
    print(f"{name:22s} {F} frames: {best:7.2f} ms = {best * 1e6 / F:6.1f} ns/frame; voiced top {np.mean(top[:, 0, 0] > 0) * 100:5.1f} %, candidates/frame {cnt.mean():6.1f}", flush=True)
    d.free()
