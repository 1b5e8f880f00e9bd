"""numpy model of the round-6 LPC path (k_lpc.hip levinson_rows_kernel_t's conditioning probe, k_lpc_exact.hip's double-double
redo), for tests/test_lpc_exact_math.py: the same formulas, vectorised over frames.  Test infrastructure only."""
import numpy as np

EPS = 2.220446049250313e-16
PROBE_EPS = 16.0 * EPS                 # LPC_PROBE_EPS (vbx_kernels.hpp)
PROBE_TOL = 1e-6                       # LPC_PROBE_TOL
NPAT = 4                               # LPC_PROBE_NPAT (k_lpc.hip)
_SPLIT = 134217729.0                   # 2^27 + 1 (Veltkamp)


def parity_metric(v, ref):
    """max_j |v_j - ref_j| / max(|ref_j|, 1e-6 max|ref|), per row"""
    ref = np.asarray(ref)
    return np.max(np.abs(v - ref) / np.maximum(np.abs(ref), 1e-6 * np.max(np.abs(ref), axis=1, keepdims=True)), axis=1).astype(np.float64)


def levinson(r, dtype=np.float64):
    """src/spectrum.rs:63-84 on rows r [F, p + 1]"""
    r = r.astype(dtype)
    F, p1 = r.shape
    a = np.zeros_like(r); a[:, 0] = 1
    err = r[:, 0].copy()
    with np.errstate(all="ignore"):
        for i in range(1, p1):
            acc = r[:, i].copy()
            for j in range(1, i):
                acc = acc + a[:, j] * r[:, i - j]
            k = -acc / err
            t = a.copy(); a[:, i] = k
            for j in range(1, i):
                a[:, j] = t[:, j] + k * t[:, i - j]
            err = err * (1 - k * k)
    return a


def lag_sums_reference_fold(xw, p):
    """src/periodic.rs:279-288: r[k] = x[0] + sum_{i >= 1} x[i] x[i + k], summed left to right in f64"""
    F, n = xw.shape
    r = np.empty((F, p + 1))
    for k in range(p + 1):
        acc = xw[:, 0].copy()
        for i in range(1, n - k):
            acc = acc + xw[:, i] * xw[:, i + k]
        r[:, k] = acc
    return r


def probe_hash(pat):
    return ((pat * 0x9E3779B1) ^ (pat << 7)) & 0xFFFFFFFF      # lpc_probe_hash


def probe(r):
    """rows listed by the conditioning probe: the f64 recursion on r and on r moved by +-PROBE_EPS |r[0]| (NPAT sign patterns);
    listed = some pattern moves the row by more than PROBE_TOL in the parity metric.  Returns (a0, listed)."""
    a0 = levinson(r)
    p1 = r.shape[1]
    d = PROBE_EPS * np.abs(r[:, :1])
    amax = np.maximum(1.0, np.max(np.abs(a0[:, 1:]), axis=1, keepdims=True))
    lim = PROBE_TOL * np.maximum(np.abs(a0), 1e-6 * amax)
    listed = np.zeros(r.shape[0], dtype=bool)
    for pat in range(1, NPAT + 1):
        h = probe_hash(pat)
        sign = np.array([1.0 if (h >> (k & 31)) & 1 else -1.0 for k in range(p1)])
        ap = levinson(r + sign[None, :] * d)
        with np.errstate(invalid="ignore"):
            listed |= np.any(np.abs(ap[:, 1:] - a0[:, 1:]) > lim[:, 1:], axis=1)
    return a0, listed


# ---- double-double arithmetic on float64 arrays (the primitives of k_lpc_exact.hip, without an FMA: Dekker's product) ----
def two_sum(a, b):
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


def fast_two_sum(a, b):
    s = a + b
    return s, b - (s - a)


def _split(a):
    t = _SPLIT * a
    hi = t - (t - a)
    return hi, a - hi


def two_prod(a, b):
    p = a * b
    ah, al = _split(a); bh, bl = _split(b)
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


def dd_add(x, y):
    sh, sl = two_sum(x[0], y[0])
    th, tl = two_sum(x[1], y[1])
    sl = sl + th
    sh, sl = fast_two_sum(sh, sl)
    sl = sl + tl
    return fast_two_sum(sh, sl)


def dd_mul(x, y):
    ph, pl = two_prod(x[0], y[0])
    pl = pl + (x[0] * y[1] + x[1] * y[0])
    return fast_two_sum(ph, pl)


def dd_neg(x):
    return -x[0], -x[1]


def dd_div(x, y):
    yi = 1.0 / y[0]
    q1 = x[0] * yi
    r = dd_add(x, dd_neg(dd_mul(y, (q1, np.zeros_like(q1)))))
    q2 = r[0] * yi
    r = dd_add(r, dd_neg(dd_mul(y, (q2, np.zeros_like(q2)))))
    q3 = r[0] * yi
    q = fast_two_sum(q1, q2)
    return dd_add(q, (q3, np.zeros_like(q3)))


def lag_sums_dd(xw, p, lanes=64):
    """Dot2 per lane segment, then the lanes' partial sums, then the fold's seed x[0] (k_lpc_exact.hip's order of operations)"""
    F, n = xw.shape
    seg = (n + lanes - 1) // lanes
    xp = np.concatenate([xw, np.zeros((F, p + 2 + seg * lanes - n + 1))], axis=1)
    out_h = np.empty((F, p + 1)); out_l = np.empty((F, p + 1))
    for k in range(p + 1):
        H = np.zeros((F, lanes)); L = np.zeros((F, lanes))
        for j in range(seg):
            i = 1 + np.arange(lanes) * seg + j
            ok = i < n
            a = np.where(ok[None, :], xp[:, np.minimum(i, n)], 0.0)
            b = np.where(ok[None, :], xp[:, np.minimum(i, n) + k], 0.0)
            ph, pl = two_prod(a, b)
            H, e = two_sum(H, ph)
            L = L + (e + pl)
        h = np.zeros(F); l = np.zeros(F)
        for u in range(lanes):
            h, e = two_sum(h, H[:, u])
            l = l + (e + L[:, u])
        h, e = two_sum(h, xw[:, 0])
        out_h[:, k], out_l[:, k] = fast_two_sum(h, l + e)
    return out_h, out_l


def levinson_dd(rh, rl):
    F, p1 = rh.shape
    one = (np.ones(F), np.zeros(F)); zero = (np.zeros(F), np.zeros(F))
    r = [(rh[:, k], rl[:, k]) for k in range(p1)]
    a = [one] + [zero] * (p1 - 1)
    err = r[0]
    for i in range(1, p1):
        acc = r[i]
        for j in range(1, i):
            acc = dd_add(acc, dd_mul(a[j], r[i - j]))
        k = dd_div(dd_neg(acc), err)
        t = list(a)
        a[i] = k
        for j in range(1, i):
            a[j] = dd_add(t[j], dd_mul(k, t[i - j]))
        err = dd_mul(err, dd_add(one, dd_neg(dd_mul(k, k))))
    return np.stack([x[0] + x[1] for x in a], axis=1)


def lpc_rows(xw, p):
    """what the library returns for frames xw [F, n]: the f64 row, or the double-double row where the probe lists the frame"""
    r = lag_sums_reference_fold(xw, p) if xw.shape[1] <= 256 else np.stack(
        [xw[:, 0] + np.sum(xw[:, 1:xw.shape[1] - k] * xw[:, 1 + k:], axis=1) for k in range(p + 1)], axis=1)
    a0, listed = probe(r)
    out = a0.copy()
    if listed.any():
        rh, rl = lag_sums_dd(xw[listed], p)
        out[listed] = levinson_dd(rh, rl)
    return out, listed
