"""numpy model of the one-pass Burg recursion of vox_box.rs_amd/csrc/k_burg_fast.hip (test infrastructure).

The reference's LPC::lpc_praat_mut (src/spectrum.rs:101-146) sums over its forward / backward error arrays once per
order; the kernel gets the same reflection coefficients from the frame's lag sums c[0..p] and its first and last p + 1
samples (the derivation is in the kernel file's header).  This model follows the kernel's recursion and its guard
statement for statement, batched over frames, so that tests can pin the MATH on the CPU against the oracle's direct
recursion -- tests/test_burg_one_pass_math.py -- before the GPU test compares the kernel with the direct kernel.
"""
import numpy as np

KAPPA_EPS = 64.0 * 2.220446049250313e-16      # k_burg_fast.hip: BF_KAPPA_EPS
TARGET = 5e-7                                 # k_burg_fast.hip: BF_TARGET


def burg_one_pass(X, P):
    """X: [F, N] windowed frames.  Returns (coeffs [F, P] in the reference's sign, trusted [F])."""
    X = np.asarray(X, dtype=np.float64)
    F, N = X.shape
    c = np.stack([np.einsum("fn,fn->f", X[:, d:], X[:, :N - d]) for d in range(P + 1)], axis=1)
    hd = X[:, :P + 1]                          # hd[k] = x[k]
    tl = X[:, ::-1][:, :P + 1]                 # tl[k] = x[N-1-k]
    L = P + 2
    a = np.zeros((F, P + 1)); a[:, 0] = 1.0
    U = np.zeros((F, L)); V = np.zeros((F, L)); rho = np.zeros((F, L)); sig = np.zeros((F, L))
    head = c[:, 0] - hd[:, 0] ** 2
    tail = c[:, 0] - tl[:, 0] ** 2
    U[:, 0], U[:, 1] = head, c[:, 1]
    V[:, 0], V[:, 1] = c[:, 1], tail
    rho[:, 0], rho[:, 1] = tail, c[:, 1]
    sig[:, 0], sig[:, 1] = head, c[:, 1]
    ok = np.ones(F, dtype=bool)
    kappa = np.zeros(F)
    with np.errstate(all="ignore"):
        for i in range(P):
            q = np.arange(1, i + 2)
            num = np.sum(a[:, i + 1 - q] * U[:, q], axis=1)
            den = np.sum(a[:, i + 1 - q] * V[:, q], axis=1) + np.sum(a[:, :i + 1] * U[:, :i + 1], axis=1)
            s1 = np.sum(np.abs(a[:, :i + 1]), axis=1)
            ok &= den > 0.0
            kappa = np.fmax(kappa, c[:, 0] * s1 * s1 / den)
            mu = 2.0 * num / den
            k = np.arange(1, i + 2)
            an = a.copy()
            an[:, k] = a[:, k] - mu[:, None] * a[:, i + 1 - k]          # a[i+1] = 0 before: an[i+1] = -mu
            a = an
            if i + 1 < P:
                q = np.arange(0, i + 2)
                fE = np.sum(hd[:, i + 1 - q] * a[:, q], axis=1)
                bE = np.sum(tl[:, q] * a[:, i + 1 - q], axis=1)
                Un = U.copy(); Vn = V.copy()
                Un[:, q] = U[:, q] - mu[:, None] * V[:, q] - hd[:, i + 1 - q] * fE[:, None]
                Vn[:, q + 1] = V[:, q] - mu[:, None] * U[:, q] - tl[:, q] * bE[:, None]
                rho[:, q] = rho[:, q] - tl[:, [i + 1]] * tl[:, i + 1 - q]
                sig[:, q] = sig[:, q] - hd[:, [i + 1]] * hd[:, i + 1 - q]
                rho[:, i + 2] = c[:, i + 2]
                sig[:, i + 2] = c[:, i + 2]
                Un[:, i + 2] = np.sum(a[:, q] * rho[:, i + 2 - q], axis=1)
                Vn[:, 0] = np.sum(a[:, q] * sig[:, i + 2 - q], axis=1)
                U, V = Un, Vn
        co = a[:, 1:]
        big = np.max(np.abs(co), axis=1)
        small = np.min(np.abs(co), axis=1)
        floor_j = np.fmax(small, 1e-6 * big)
        trusted = ok & (KAPPA_EPS * kappa * big <= TARGET * floor_j)
    return co, trusted


def parity_metric(got, exp):
    """per row: max_j |got - exp| / max(|exp_j|, 1e-6 max|exp|)  (tests/conftest.py rel_close, as a number)"""
    sc = np.max(np.abs(exp), axis=1, keepdims=True)
    with np.errstate(all="ignore"):
        return np.max(np.abs(got - exp) / np.maximum(np.abs(exp), 1e-6 * sc + 1e-300), axis=1)


def adversarial_frames(N, rng, count=60):
    """Frames the autocorrelation-domain recursion cannot be trusted on (or only just): pure tones with noise floors from
    1e-1 down to 1e-9, sums of tones, resonators with poles up to 1e-4 from the unit circle, DC, quantised tones, white
    noise at scales from 1e-6 to 1e2, silence, an impulse, a frame with a NaN."""
    n = np.arange(N)
    out = []
    for t in range(count):
        kind = t % 6
        noise = 10.0 ** (-rng.uniform(1, 9))
        if kind == 0:
            x = np.sin(2 * np.pi * rng.uniform(0.005, 0.4) * n + rng.uniform(0, 6)) + noise * rng.standard_normal(N)
        elif kind == 1:
            x = sum(rng.uniform(0.1, 1) * np.sin(2 * np.pi * rng.uniform(0.005, 0.45) * n + rng.uniform(0, 6)) for _ in range(3))
            x = x + noise * rng.standard_normal(N)
        elif kind == 2:
            r = 1 - 10.0 ** (-rng.uniform(1, 4)); th = rng.uniform(0.05, 3)
            e = rng.standard_normal(N + 200); y = np.zeros(N + 200)
            for k in range(2, N + 200):
                y[k] = e[k] + 2 * r * np.cos(th) * y[k - 1] - r * r * y[k - 2]
            x = y[200:] / np.max(np.abs(y[200:]))
        elif kind == 3:
            x = np.ones(N) * rng.uniform(0.1, 1) + noise * rng.standard_normal(N)
        elif kind == 4:
            x = np.round(np.sin(2 * np.pi * rng.uniform(0.005, 0.1) * n) * rng.uniform(3, 300)) / 300.0
        else:
            x = rng.standard_normal(N) * 10.0 ** rng.uniform(-6, 2)
        out.append(x)
    out.append(np.zeros(N))                                   # silence: Err(LPC)
    imp = np.zeros(N); imp[N // 2] = 1.0
    out.append(imp)
    bad = rng.standard_normal(N); bad[N // 3] = np.nan
    out.append(bad)
    return np.array(out)
