# per-frame survivors of the bound passes, voiced vs unvoiced frames (CPU experiment, oracle only)
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
o = g.load_oracle(); pkg = g.load_package()
import importlib
synth = importlib.import_module(g.PKG_NAME + ".synth")
N, H, SR = 1200, 480, 48000.0
audio = synth.synth_speech(10 * 48000 + N, sample_offset=0)
w = o.window("hanning", N); lw = o.window("hanning_lag", N)
golden = 1. - 0.6180339887498948482045868343656381177203091798057628621
B = 4
def bound(y, ay_cs, offset, nx, x, heads):
    nl = int(np.floor(x)); nr = nl + 1; phil = x - nl; phir = 1 - phil
    D = 1200
    if offset + nr < D: D = max(offset + nr, 0)
    if offset + nl + D >= nx: D = nx - offset + nl - 1
    n = np.arange(D + 1)
    aL = np.pi * (phil + n); iL = np.maximum(offset + nr - n, 0)
    cLs = np.sin(aL) / aL * (0.5 + 0.5 * np.cos(aL / (phil + D)))
    aR = np.pi * (phir + n); iR = np.minimum(offset + nl + n, len(y) - 1)
    cRs = np.sin(aR) / aR * (0.5 + 0.5 * np.cos(aR / (phir + D)))
    tL = y[iL] * cLs; tR = y[iR] * cRs
    fv = tL.sum() + tR.sum()
    def rs(i0, i1):
        i0 = (i0 // B) * B; i1 = ((i1 // B) + 1) * B - 1
        i0 = max(i0, 0); i1 = min(i1, len(y) - 1)
        return ay_cs[i1 + 1] - ay_cs[i0] if i1 >= i0 else 0.0
    out = []
    for Dh in heads:
        part = tL[:Dh].sum() + tR[:Dh].sum(); tail = 0.0; lo = Dh
        while lo < len(tL):
            hi = min(2 * lo, len(tL))
            tail += rs(iL[hi - 1], iL[lo]) * abs(cLs[lo]) + rs(iR[lo], iR[hi - 1]) * abs(cRs[lo])
            lo = hi
        out.append(min(part + tail, 1.0))
    return min(fv, 1.0), out
heads = (8, 32, 128)
stats = {"voiced": [], "unvoiced": []}
for t in range(0, 1000, 12):
    x = audio[t * H:t * H + N] * w
    st, cands, cnt = o.pitch(x, SR, 0.2, 75., 600.)
    kind = "unvoiced" if cands[0, 0] == 0.0 else "voiced"
    r = o.autocorrelate(x, N); r = r / np.max(np.abs(r)); yv = r / lw
    y = np.concatenate([yv, np.zeros(N)]); cs = np.concatenate([[0], np.cumsum(np.abs(y))])
    b = N // 2; offset = -b - 1; nx = b - offset
    rows = []
    for k in range(1, b - 1):
        if yv[k - 1] < yv[k] > yv[k + 1]:
            dr = 0.5 * (yv[k + 1] - yv[k - 1]); d2r = 2 * yv[k] - (yv[k - 1] - yv[k + 1])
            freq = SR / (k + dr / d2r)
            if not (75. < freq < 600.): continue
            nn = SR / freq - offset
            rows.append(bound(y, cs, offset, nx, (nn - 1) + golden * 2, heads))
    if not rows: continue
    # champion = best head-8 bound; bar after champion = max(threshold, final top strength)  (kmax = 1)
    bar = max(0.2, cands[0, 1])
    fvs = np.array([r_[0] for r_ in rows]); ubs = np.array([r_[1] for r_ in rows])
    stats[kind].append((len(rows), int((fvs >= bar).sum()), *[int((ubs[:, i] >= bar).sum()) for i in range(len(heads))]))
for kind, v in stats.items():
    a = np.array(v, dtype=float)
    print(kind, "frames", len(v), "mean: candidates %.1f, exact f(v0) >= bar %.2f," % (a[:, 0].mean(), a[:, 1].mean()),
          ", ".join("head %d survivors %.2f" % (h, a[:, 2 + i].mean()) for i, h in enumerate(heads)))
