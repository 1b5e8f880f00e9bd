# how much would a truncated first evaluation + rigorous tail bound prune?  (CPU experiment, oracle only)
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
def terms(y, offset, nx, x):
    nl = int(np.floor(x)); nr = nl + 1; phil = x - nl; phir = 1 - phil
    D = 1200
    if offset + nr < D: D = max(offset + nr, 0)
    if offset + nl + D >= nx: D = nx - offset + nl - 1
    n = np.arange(D + 1)
    aL = np.pi * (phil + n); iL = np.maximum(offset + nr - n, 0)
    tL = y[iL] * np.sin(aL) / aL * (0.5 + 0.5 * np.cos(aL / (phil + D)))
    cL = np.abs(np.sin(aL) / aL * (0.5 + 0.5 * np.cos(aL / (phil + D))))
    aR = np.pi * (phir + n); iR = np.minimum(offset + nl + n, len(y) - 1)
    tR = y[iR] * np.sin(aR) / aR * (0.5 + 0.5 * np.cos(aR / (phir + D)))
    cR = np.abs(np.sin(aR) / aR * (0.5 + 0.5 * np.cos(aR / (phir + D))))
    return tL, tR, np.abs(y[iL]), np.abs(y[iR]), cL, cR, iL, iR
tot = {}; ncand_tot = 0; exact_pruned = 0
for t in range(0, 1000, 25):
    x = audio[t * H:t * H + N] * w
    st, cands, cnt = o.pitch(x, SR, 0.2, 75., 600.)
    bar = cands[0, 1]
    r = o.autocorrelate(x, N); r = r / np.max(np.abs(r)); yv = r / lw
    y = np.concatenate([yv, np.zeros(N)])
    b = N // 2; offset = -b - 1; nx = b - offset
    for k in range(1, b - 1):
        if yv[k - 1] < yv[k] > yv[k + 1]:
            dr = 0.5 * (yv[k + 1] - yv[k - 1]); d2r = 2 * yv[k] - yv[k - 1] - yv[k + 1]
            freq = SR / (k + dr / d2r)
            if not (75. < freq < 600.): continue
            nn = SR / freq - offset
            v0 = (nn - 1) + golden * 2
            tL, tR, aL, aR, cL, cR, iL, iR = terms(y, offset, nx, v0); ay = np.abs(y); cs = np.concatenate([[0], np.cumsum(ay)])
            fv = tL.sum() + tR.sum(); ub = min(fv, 1.0)
            ncand_tot += 1
            if ub < bar: exact_pruned += 1
            for D in (8, 12, 16):
              for B in (1, 4, 16):
                part = tL[:D].sum() + tR[:D].sum()
                tail = 0.0; lo = D
                def rs(i0, i1):
                    i0 = (i0 // B) * B; i1 = ((i1 // B) + 1) * B - 1
                    i0 = max(i0, 0); i1 = min(i1, len(ay) - 1)
                    return cs[i1 + 1] - cs[i0] if i1 >= i0 else 0.0
                while lo < len(tL):
                    hi = min(2 * lo, len(tL))
                    tail += rs(iL[hi - 1], iL[lo]) * cL[lo] + rs(iR[lo], iR[hi - 1]) * cR[lo]
                    lo = hi
                tot.setdefault((D, B), [0, 0.0]); tot[(D, B)][1] += tail
                if min(part + tail, 1.0) < bar: tot[(D, B)][0] += 1
print("candidates", ncand_tot, "pruned by exact first eval", exact_pruned)
for D, (c, tl) in sorted(tot.items()): print("D", D, "pruned", c, "mean tail bound", tl / ncand_tot)
