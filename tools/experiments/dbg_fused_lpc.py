#!/usr/bin/env python3
"""Where does the fused kernel's LPC (Levinson on the FFT's r[0..12]) leave the 1e-6 band around the oracle's?  For the
frames outside it: the direct-sum entry point (vbx_autocorr_lpc_f64), the oracle, and the same recursion in long double on
long-double lag sums -- i.e. how far the ORACLE itself is from the exact answer (conditioning).
usage: python3 tools/experiments/dbg_fused_lpc.py [n_frames=60000] [frame_len=1200] [hop=480]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg, o = g.load_package(), g.load_oracle()
vb = pkg.VoxBox(0)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
H = int(sys.argv[3]) if len(sys.argv) > 3 else 480
SR, P = 48000.0, 12
ns = (nf - 1) * H + N
audio_d = vb.synth_speech(ns, sample_offset=5 * 48000)
han = vb.window(pkg.WINDOW_HANNING, N)
params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=0, mfcc=None)
l0, ln = params.columns()["lpc"]
rec, st = vb.analyze_frames(audio_d, params, frame_len=N, stride=H, n_frames=nf)
r_d, a_d = vb.autocorr_lpc(audio_d, P, frame_len=N, stride=H, n_frames=nf, window=han)
wh = o.window("hanning", N)


def lev_ld(r):
    r = np.asarray(r, dtype=np.longdouble)
    a = np.zeros(P + 1, dtype=np.longdouble); a[0] = 1; err = r[0]
    for i in range(1, P + 1):
        acc = r[i] + sum(a[j] * r[i - j] for j in range(1, i))
        k = -acc / err
        t = a.copy(); a[i] = k
        for j in range(1, i): a[j] = t[j] + k * t[i - j]
        err = err * (1 - k * k)
    return a, err


def close(a, b):
    return np.all(np.abs(a - b) <= 1e-6 * np.maximum(np.abs(b), 1e-6 * np.max(np.abs(b))))


# cheap screen with the direct-sum GPU result, then the oracle on the suspects and a sample
sus = [t for t in range(nf) if not close(rec[t, l0:l0 + ln], a_d[t])]
print("frames where fused and direct-sum GPU LPC differ beyond 1e-6:", len(sus), "of", nf)
shown = 0
for t in sus[:8]:
    x = audio_d.numpy_slice(t * H, N) * wh
    el = o.lpc(o.autocorrelate(x, P + 1), P)
    xl = x.astype(np.longdouble)
    rl = np.array([xl[0] + np.sum(xl[1:N - k] * xl[1 + k:N]) for k in range(P + 1)])
    al, err = lev_ld(rl)
    dev = lambda v: float(np.max(np.abs(v - al) / np.maximum(np.abs(al), 1e-6 * np.max(np.abs(al)))))
    print(f"frame {t}: prediction error / r0 = {float(err / rl[0]):.3e}; max rel. deviation from the long-double answer: "
          f"oracle {dev(el):.2e}, GPU direct sums {dev(a_d[t]):.2e}, GPU fused (FFT) {dev(rec[t, l0:l0 + ln]):.2e}")
