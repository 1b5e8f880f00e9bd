"""numpy model of vox_box.rs_amd/csrc/k_roots_fast.hip (test infrastructure): the resonances of find_formants from the Burg
coefficients by one Laguerre solve per conjugate PAIR of the real polynomial rev([1, a1..ap]) -- three real synthetic
divisions per evaluation, converged iteration with the deflated polynomial's own degree, deflation by the real quadratic,
one Newton step on the original polynomial as polish and check.  Frames are batched along axis 0 as the kernel batches
them across lanes; a frame the kernel would do again by the reference's iteration is returned as flagged.

The reference's method (src/polynomial.rs:34-152: complex Laguerre from (-2, -2) with the degree fixed, always 20 iterations,
deflation by single complex roots) is oracle/vbx_oracle.c's vbxo_find_roots_mut; tests/test_roots_fast_math.py holds this
model to it through the sorted resonance rows, which is all that find_formants lets out (src/lib.rs:94-110)."""
import numpy as np

MAX_IT, CONV, CHECK, REAL = 32, 1e-7, 1e-7, 1e-9
START = (0.8, 0.4)
FRACTIONS = (0.5, 0.25, 0.75, 0.13, 0.38, 0.62, 0.88)


def eval3(c, top, x, y):
    """p, p', p'' at x + iy and the first division's coefficients b, for real polynomials c[:, 0..top] (index = power)."""
    r = 2 * x; s = x * x + y * y
    F = c.shape[0]
    b = np.zeros((F, top + 3)); e = np.zeros((F, top + 3)); g = np.zeros((F, top + 3))
    for k in range(top, -1, -1):
        b[:, k] = c[:, k] + r * b[:, k + 1] - s * b[:, k + 2]
    for k in range(top - 2, -1, -1):
        e[:, k] = b[:, k + 2] + r * e[:, k + 1] - s * e[:, k + 2]
    for k in range(top - 4, -1, -1):
        g[:, k] = e[:, k + 2] + r * g[:, k + 1] - s * g[:, k + 2]
    p = (b[:, 0] - x * b[:, 1]) + 1j * (y * b[:, 1])
    Q = (e[:, 0] - x * e[:, 1]) + 1j * (y * e[:, 1])
    Q2 = (g[:, 0] - x * g[:, 1]) + 1j * (y * g[:, 1])
    dp = 2j * y * Q + b[:, 1]
    ddp = 2 * Q - 8 * y * y * Q2 + 4j * y * e[:, 1]
    return p, dp, ddp, b


def resonance_rows(coeffs, sample_rate=48000.0, n_slots=32):
    """coeffs: [F, P] Burg coefficients.  Returns (rows [F, n_slots, 2] sorted by frequency and zero padded, count [F],
    status [F], flagged [F])."""
    a = np.asarray(coeffs, dtype=np.float64)
    F, P = a.shape
    c0 = np.concatenate([a[:, ::-1], np.ones((F, 1))], axis=1)
    status = np.where(c0[:, 0] == 0.0, 4, 0).astype(np.int32)
    flagged = (status == 0) & ~np.all(np.isfinite(c0), axis=1)
    m = np.where((status == 0) & ~flagged, P, 0)
    c = c0.copy()
    roots = [[] for _ in range(F)]

    def emit(idx, x, y):
        p, dp, _, _ = eval3(c0[idx], P, x, y)
        with np.errstate(all="ignore"):
            step = p / dp
            bad = ~(np.abs(step) <= CHECK * np.hypot(x, y))
        z = (x + 1j * y) - step
        for j, i in enumerate(idx):
            if bad[j]:
                flagged[i] = True
            roots[i].append(z[j])

    with np.errstate(all="ignore"):
        while np.any(m > 2):
            top = int(m.max())
            sel = np.nonzero(m > 2)[0]
            cc = c[sel]; n = m[sel].astype(np.float64)
            x = np.full(sel.size, START[0]); y = np.full(sel.size, START[1])
            done = np.zeros(sel.size, dtype=bool)
            for it in range(MAX_IT):
                p, dp, ddp, _ = eval3(cc, top, x, y)
                G = dp / p; H = G * G - ddp / p
                sq = np.sqrt((n - 1) * (n * H - G * G))
                d1, d2 = G + sq, G - sq
                dz = n / np.where(np.abs(d1) > np.abs(d2), d1, d2)
                frac = it >= 8 and (it - 8) % 5 == 0
                if frac:
                    dz = dz * FRACTIONS[((it - 8) // 5) % 7]
                exact = p == 0
                upd = ~done & ~exact
                x = np.where(upd, x - dz.real, x); y = np.where(upd, y - dz.imag, y)
                done |= exact | ((not frac) & (np.abs(dz) <= CONV * np.hypot(x, y)))
                if done.all():
                    break
            flagged[sel[~done]] = True
            m[sel[~done]] = 0
            ok = sel[done]; x, y, cc = x[done], np.abs(y[done]), cc[done]
            real = np.abs(y) <= REAL * np.hypot(x, y)
            if np.any(~real):
                emit(ok[~real], x[~real], y[~real])
                _, _, _, b = eval3(cc[~real], top, x[~real], y[~real])
                nb = np.zeros((int(np.sum(~real)), P + 1)); nb[:, :top - 1] = b[:, 2:top + 1]
                c[ok[~real]] = nb; m[ok[~real]] -= 2
            if np.any(real):
                q = np.zeros((int(np.sum(real)), P + 1)); t = np.zeros(int(np.sum(real)))
                for k in range(P, -1, -1):
                    ck = cc[real][:, k]; q[:, k] = t; t = x[real] * t + ck
                c[ok[real]] = q; m[ok[real]] -= 1
        last = np.nonzero(m == 2)[0]
        if last.size:
            a2, a1, a0 = c[last, 2], c[last, 1], c[last, 0]
            disc = a1 * a1 - 4 * a2 * a0
            neg = disc < 0
            flagged[last[~neg & ~(disc >= 0)]] = True
            if np.any(neg):
                emit(last[neg], -a1[neg] / (2 * a2[neg]), np.abs(np.sqrt(-disc[neg]) / (2 * a2[neg])))
    rows = np.zeros((F, n_slots, 2)); count = np.zeros(F, dtype=np.int32)
    fm = sample_rate / (2 * np.pi)
    for i in range(F):
        if status[i] != 0 or flagged[i]:
            continue
        res = []
        for z in roots[i]:
            if not z.imag > 0.0:
                continue
            r, th = abs(z), np.angle(z)
            if r > 1.0:
                zi = 1.0 / np.conj(z); r, th = abs(zi), np.angle(zi)
            fr, bw = fm * th, -2.0 * fm * np.log(r)
            if 50.0 < fr < sample_rate * 0.5 - 50.0:
                res.append((fr, bw))
        res.sort(key=lambda t: t[0])
        count[i] = len(res)
        for j, t in enumerate(res):
            rows[i, j] = t
    return rows, count, status, flagged
