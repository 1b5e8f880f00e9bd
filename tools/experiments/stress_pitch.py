# one-off stress of the pitch path: odd signals, kmax heads consistent, oracle agreement on samples
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0); o = g.load_oracle()
N, SR = 1200, 48000.0
rng = np.random.default_rng(123)
t = np.arange(N) / SR
w = o.window("hanning", N)
frames = []
for i in range(400):
    kind = i % 10
    if kind == 0: x = rng.standard_normal(N)
    elif kind == 1: x = np.sin(2*np.pi*rng.uniform(60, 700)*t + rng.uniform(0, 6))
    elif kind == 2: x = np.sign(np.sin(2*np.pi*rng.uniform(80, 400)*t))
    elif kind == 3: x = np.sin(2*np.pi*(100 + 3000*t)*t)
    elif kind == 4: x = np.zeros(N); x[rng.integers(0, N, 5)] = 1.0
    elif kind == 5: x = 0.5 + 0.01*rng.standard_normal(N)
    elif kind == 6: x = 1e-150*np.sin(2*np.pi*200*t) 
    elif kind == 7: x = 1e120*np.sin(2*np.pi*150*t)
    elif kind == 8: x = np.sin(2*np.pi*120*t)*(1+0.5*np.sin(2*np.pi*7*t)) + 0.2*rng.standard_normal(N)
    else: x = np.clip(3*np.sin(2*np.pi*rng.uniform(75, 600)*t), -1, 1)
    frames.append(x * w)
X = np.array(frames)
res = {}
for kmax in (1, 3, 64):
    res[kmax] = vb.pitch(X, SR, 0.2, 75., 600., kmax=kmax)
c64 = res[64][0]
for kmax in (1, 3):
    assert np.array_equal(res[kmax][1], res[64][1]) and np.array_equal(res[kmax][2], res[64][2])
    assert np.array_equal(res[kmax][0], c64[:, :kmax]), kmax
bad = 0
for f in range(X.shape[0]):
    es, ec, en = o.pitch(X[f], SR, 0.2, 75., 600.)
    st, cnt = res[1][2][f], res[1][1][f]
    assert st == es and cnt == (en if es == 0 else 0), (f, st, es, cnt, en)
    if es == 0:
        tie = en > 1 and abs(ec[0, 1] - ec[1, 1]) < 1e-3
        ok = abs(res[1][0][f, 0, 0] - ec[0, 0]) <= 1e-4*abs(ec[0, 0]) and abs(res[1][0][f, 0, 1] - ec[0, 1]) <= 1e-4
        bad += 0 if (ok or tie) else 1
print("frames", X.shape[0], "status histogram", np.bincount(res[1][2], minlength=5), "top mismatches", bad)
