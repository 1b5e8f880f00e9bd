"""The resonance kernel of find_formants from conjugate pairs (k_roots_fast.hip) against the reference's own iteration on
every frame (VBX_ROOTS_DIRECT=1: k_roots.hip, which follows src/polynomial.rs:34-152 operation by operation), against
its numpy model, and against the oracle."""
import numpy as np
import pytest

from roots_fast_model import resonance_rows

pytestmark = pytest.mark.gpu
P = 12
SR = 48000.0


def _both(vb, monkeypatch, fn):
    monkeypatch.delenv("VBX_ROOTS_DIRECT", raising=False)
    a = fn()
    redone = vb.last_roots_direct_count()
    monkeypatch.setenv("VBX_ROOTS_DIRECT", "1")
    b = fn()
    assert vb.last_roots_direct_count() == -1
    monkeypatch.delenv("VBX_ROOTS_DIRECT", raising=False)
    return a, b, redone


def _rel(a, b):
    nz = b != 0
    assert np.all(a[~nz] == 0)
    return float(np.max(np.abs(a[nz] - b[nz]) / np.abs(b[nz]))) if np.any(nz) else 0.0


@pytest.mark.parametrize("n,hop", [(512, 512), (1200, 480), (2048, 1024)])
def test_resonance_rows_equal_the_reference_iterations(vb, pkg, oracle, monkeypatch, n, hop):
    """60,000 speech frames: statuses and counts equal, every frequency and bandwidth within 1e-8 relative of the
    reference-faithful kernel's (the gate against the oracle is 1e-4), the tracks within 1e-8; against the oracle's own
    walk on a sample of the frames: counts equal, rows within 1e-8."""
    F = 60000
    audio = vb.synth_speech((F - 1) * hop + n, sample_offset=17 * 48000)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    seg = np.arange(0, F, 500, dtype=np.int64)
    run = lambda: vb.find_formants(audio, SR, P, est0, seg_start=seg, frame_len=n, stride=hop, n_frames=F)
    a, b, redone = _both(vb, monkeypatch, run)
    assert 0 <= redone <= F // 1000, redone
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["count"], b["count"])
    assert _rel(a["res"], b["res"]) <= 1e-8
    assert np.max(np.abs(a["formants"] - b["formants"]) / np.abs(b["formants"])) <= 1e-8
    host = audio.numpy()
    s = oracle.soak(host, n, hop, 0, 3000, P, SR, oracle.SOAK_FORMANTS)
    assert np.array_equal(a["status"][:3000], s["ff_status"]) and np.array_equal(a["count"][:3000], s["res_count"])
    assert _rel(a["res"][:3000], s["res"]) <= 1e-8
    audio.free()


@pytest.mark.parametrize("order", [8, 10, 13, 14, 16])
def test_the_other_orders(vb, pkg, oracle, monkeypatch, order):
    """Orders 10 and 13 are what the reference's own callers pass (tests/lib.rs:23,52); an odd order always has a real root."""
    F, n, hop = 20000, 512, 256
    audio = vb.synth_speech((F - 1) * hop + n, sample_offset=29 * 48000)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    seg = np.arange(0, F, 500, dtype=np.int64)
    run = lambda: vb.find_formants(audio, SR, order, est0, seg_start=seg, frame_len=n, stride=hop, n_frames=F)
    a, b, redone = _both(vb, monkeypatch, run)
    assert 0 <= redone <= F // 1000, redone
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["count"], b["count"])
    assert _rel(a["res"], b["res"]) <= 1e-8
    assert np.max(np.abs(a["formants"] - b["formants"]) / np.abs(b["formants"])) <= 1e-8
    s = oracle.soak(audio.numpy(), n, hop, 0, 2000, order, SR, oracle.SOAK_FORMANTS)
    assert np.array_equal(a["status"][:2000], s["ff_status"]) and np.array_equal(a["count"][:2000], s["res_count"])
    assert _rel(a["res"][:2000], s["res"]) <= 1e-8
    audio.free()
    # an order without an instantiation takes the reference's iteration
    vb.find_formants(vb.synth_speech(20 * 512).numpy().reshape(20, 512), SR, 11, est0)
    assert vb.last_roots_direct_count() == -1


def _frames_from_polys(rng, kind, count, n=512):
    """Frames whose order-12 Burg polynomial is of a given kind, built by filtering noise through an all-pole filter."""
    out = []
    for _ in range(count):
        if kind == "clustered":        # three pole pairs within 1e-3 of each other, close to the unit circle
            th0 = rng.uniform(0.3, 2.5)
            poles = [0.995 * np.exp(1j * (th0 + d)) for d in (0.0, 1e-3, 2e-3)] + [rng.uniform(0.5, 0.9) * np.exp(1j * rng.uniform(0.2, 2.9)) for _ in range(3)]
        elif kind == "circle":         # evenly spaced on a circle: Laguerre's limit cycles
            r = rng.uniform(0.7, 0.95); ph = rng.uniform(0, np.pi / 6)
            poles = [r * np.exp(1j * (ph + np.pi * (2 * k + 1) / 12)) for k in range(6)]
        elif kind == "real":           # four real poles and four pairs
            poles = [rng.uniform(0.5, 0.97) * np.exp(1j * rng.uniform(0.2, 2.9)) for _ in range(4)]
            real = [rng.uniform(-0.95, 0.95) for _ in range(4)]
        else:
            poles = [rng.uniform(0.3, 0.999) * np.exp(1j * rng.uniform(0.05, 3.1)) for _ in range(6)]
        a = np.array([1.0])
        for z in poles:
            a = np.convolve(a, [1.0, -2 * z.real, abs(z) ** 2])
        if kind == "real":
            a = a[:9]
            a = np.array([1.0])
            for z in poles:
                a = np.convolve(a, [1.0, -2 * z.real, abs(z) ** 2])
            for x in real:
                a = np.convolve(a, [1.0, -x])
        e = rng.standard_normal(n + 400)
        y = np.zeros(n + 400)
        for t in range(n + 400):
            acc = e[t]
            for k in range(1, min(t, a.size - 1) + 1):
                acc -= a[k] * y[t - k]
            y[t] = acc
        y = y[400:]
        out.append(y / np.max(np.abs(y)))
    return np.array(out)


def test_hard_polynomials(vb, pkg, oracle, monkeypatch):
    """Pole clusters 1e-3 apart next to the unit circle, poles evenly spaced on a circle (the polynomials on which Laguerre
    from a symmetric start cycles), real poles, a pure tone, DC, silence and a NaN frame: statuses and counts are the
    reference iteration's; rows agree to 1e-6 relative wherever the oracle's own rows are stable under a 1e-13
    perturbation of the frame (on a cluster neither method's sixth digit means anything: the GATE is 1e-4), and the kernel
    agrees with its numpy model about which frames to hand to the reference's iteration."""
    rng = np.random.default_rng(5)
    n = 512
    x = np.concatenate([_frames_from_polys(rng, k, 24, n) for k in ("clustered", "circle", "real", "random")])
    t = np.arange(n)
    extra = np.array([np.sin(2 * np.pi * 0.03 * t), np.ones(n) * 0.5, np.zeros(n), np.full(n, np.nan)])
    x = np.concatenate([x, extra])
    F = x.shape[0]
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    run = lambda: vb.find_formants(x, SR, P, est0)
    a, b, redone = _both(vb, monkeypatch, run)
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["count"], b["count"])
    assert a["status"][F - 2] != 0 and a["count"][F - 1] == 0        # silence: Err(LPC); NaN: no resonance, no error
    # the kernel and its model decide alike (the model on the kernel's own coefficients)
    ok = a["status"] == 0
    _, _, _, flagged = resonance_rows(np.nan_to_num(a["coeffs"][ok], nan=np.nan))
    assert abs(int(np.sum(flagged)) - redone) <= 2, (int(np.sum(flagged)), redone)
    stable = np.zeros(F, dtype=bool)
    for f in range(F - 4):
        st1, _, r1, _ = oracle.find_formants(x[f], SR, P, est0)
        st2, _, r2, _ = oracle.find_formants(x[f] * (1.0 + 1e-13 * np.cos(t)), SR, P, est0)
        assert a["status"][f] == st1, f
        nz = r1 != 0
        stable[f] = st1 == 0 and st2 == 0 and np.array_equal(nz, r2 != 0) and np.all(np.abs(r1[nz] - r2[nz]) <= 1e-9 * np.abs(r1[nz]))
        if stable[f]:
            assert np.array_equal(a["res"][f] != 0, nz), f
            assert np.all(np.abs(a["res"][f][nz] - r1[nz]) <= 1e-6 * np.abs(r1[nz])), (f, a["res"][f][nz], r1[nz])
    assert stable.sum() > F // 2, stable.sum()
    assert _rel(a["res"][stable], b["res"][stable]) <= 1e-6
