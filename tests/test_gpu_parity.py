"""Parity tests proper: every entry point of libvoxbox_hip.so (called through the C ABI)
against the CPU oracle on identical inputs.  Tolerances are BASELINE.json's:
autocorrelation / LPC 1e-6 relative (SURVEY 8d metric), pitch / formant Hz 1e-4 relative,
candidate counts and statuses exact.  Needs a real MI355X: run with `-m gpu`.
"""
import os
import wave

import numpy as np
import pytest

from conftest import rel_close

pytestmark = pytest.mark.gpu

SR = 48000.0
N48, H48 = 1200, 480


def _read_wav16(path):
    with wave.open(path, "rb") as w:
        sr = w.getframerate()
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return pcm.astype(np.float64) / 32767.0, float(sr)


@pytest.fixture(scope="module")
def audio(vb):
    """6 s of device-generated synthetic speech pulled back to the host: seconds 2..8 of the
    stream, i.e. voiced glide + one fully unvoiced second (4..5)."""
    d = vb.synth_speech(6 * 48000, sample_offset=2 * 48000)
    a = d.numpy()
    d.free()
    return a


def _frames(audio, n, hop, idx):
    return np.stack([audio[t * hop:t * hop + n] for t in idx])


# ---- cross-lane helpers -----------------------------------------------------------------

def test_lane_helpers(vb):
    both = vb.selftest_lanes()
    o, gsum = both[0], both[1]
    v = 1.0 + 0.37 * np.arange(64) + 1e-3 * ((np.arange(64) * 7919) % 64)
    assert np.array_equal(o[:, 0], o[:, 1]), "DPP wave_shl:1 != __shfl_down"
    assert np.array_equal(o[:, 2], o[:, 3]), "DPP wave_shr:1 != __shfl_up"
    assert np.allclose(o[:-1, 0], v[1:], rtol=1e-14, atol=0) and o[63, 0] == 0.0
    assert np.allclose(o[1:, 2], v[:-1], rtol=1e-14, atol=0) and o[0, 2] == 0.0
    assert np.all(np.abs(o[:, 4] - v.sum()) < 1e-10) and np.all(o[:, 4] == o[0, 4])
    assert np.all(np.abs(o[:, 5] - v.sum()) < 1e-10)
    assert np.allclose(o[:, 6], v.max(), rtol=1e-14) and np.all(o[:, 6] == o[0, 6])
    assert np.allclose(o[:, 7], v[17], rtol=1e-14) and np.all(o[:, 7] == o[0, 7])
    for j, g in enumerate((4, 8, 16, 32, 64)):        # group sums: right value, bit-identical inside a group
        col = gsum[:, j].reshape(64 // g, g)
        assert np.allclose(col[:, 0], v.reshape(64 // g, g).sum(axis=1), rtol=1e-13), g
        assert np.all(col == col[:, :1]), g
    assert np.allclose(gsum[:, 5], np.cumsum(v), rtol=1e-14), "DPP inclusive scan"


def test_synth_matches_host_statement(vb, pkg, audio):
    import importlib
    import __graft_entry__ as g
    synth = importlib.import_module(g.PKG_NAME + ".synth")
    host = synth.synth_speech(audio.size, sample_offset=2 * 48000)
    assert np.max(np.abs(host - audio)) < 1e-9
    assert np.max(np.abs(audio)) < 1.0 and np.std(audio) > 0.01


# ---- autocorrelate / normalize / lpc --------------------------------------------------------

@pytest.mark.parametrize("n,lags", [(512, 13), (512, 1), (16, 16), (100, 7), (1200, 13), (1200, 1200),
                                    (333, 333), (2048, 17), (4096, 40), (640, 321), (64, 64), (5, 5),
                                    (1300, 1290), (4096, 4096), (1280, 257),      # several matrix-core passes / partial tiles
                                    # 64 lags or more of a 512..4096-sample frame: one FFT of the zero-padded frame
                                    (512, 64), (512, 511), (1024, 1024), (1103, 1103), (1200, 65), (2047, 1001), (2048, 2048),
                                    (3000, 64), (640, 63), (1024, 18), (2048, 40)])
def test_autocorrelate(vb, oracle, n, lags):
    rng = np.random.default_rng(n * 1000 + lags)
    x = rng.uniform(-1, 1, (9, n))          # rectangular frames: x[0] != 0 exercises the Q1 seed
    got = vb.autocorrelate(x, lags)
    for f in range(x.shape[0]):
        exp = oracle.autocorrelate(x[f], lags)
        assert np.all(rel_close(got[f], exp)), (f, np.max(np.abs(got[f] - exp)))


@pytest.mark.parametrize("n,lags", [(1200, 1200), (1024, 100), (2048, 2047), (777, 300), (4096, 4096)])
def test_autocorrelate_fft_and_direct_kernels_agree(pkg, oracle, audio, monkeypatch, n, lags):
    """Many lags of a long frame come from one FFT; VBX_PITCH_MFMA=1 keeps the direct (matrix-core) lag sums.  Windowed speech
    frames: the sums at the far lags are 1e-13 of r[0] -- the FFT's rounding error must stay under the tolerance's floor."""
    F = (audio.size - n) // 997 + 1
    x = _frames(audio, n, 997, range(0, F, max(1, F // 40))) * oracle.window("hanning", n)
    res = {}
    for name, var in (("fft", None), ("direct", "VBX_PITCH_MFMA")):
        if var:
            monkeypatch.setenv(var, "1")
        v = pkg.VoxBox(0)
        if var:
            monkeypatch.delenv(var)
        try:
            res[name] = v.autocorrelate(x, lags)
        finally:
            v.close()
    assert not np.array_equal(res["fft"], res["direct"])                       # two kernels really ran
    for f in range(x.shape[0]):
        assert np.all(rel_close(res["fft"][f], res["direct"][f])), f
        assert np.all(rel_close(res["fft"][f], oracle.autocorrelate(x[f], lags))), f


def test_autocorrelate_strided_windowed(vb, oracle, pkg, audio):
    F = pkg.frame_count(audio.size, N48, H48)
    d = vb.to_device(audio)
    w = vb.window(pkg.WINDOW_HANNING, N48)
    got = vb.autocorrelate(d, 13, frame_len=N48, stride=H48, n_frames=F, window=w)
    wh = oracle.window("hanning", N48)
    for f in range(0, F, 37):
        exp = oracle.autocorrelate(audio[f * H48:f * H48 + N48] * wh, 13)
        assert np.all(rel_close(got[f], exp)), f
    d.free()


def test_empty_batch_is_a_noop(vb):
    assert vb.autocorrelate(np.zeros((0, 512)), 13).shape == (0, 13)
    c, cnt, st = vb.pitch(np.zeros((0, 512)), SR, 0.2, 75., 600.)
    assert cnt.size == 0


def test_normalize(vb, oracle):
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((7, 100))
    got = vb.normalize(rows)
    for f in range(7):
        assert np.all(rel_close(got[f], oracle.normalize(rows[f]), 1e-14))


def test_lpc_kat(vb, oracle):
    """src/spectrum.rs:470-487 test_lpc through the GPU path."""
    s = oracle.sine(8, 8.0, 1.0)
    r = vb.autocorrelate(s[None, :], 8)
    auto = vb.normalize(r)
    assert np.all(np.abs(auto[0] - [1.0, 0.7071, 0.1250, -0.3536, -0.5, -0.3536, -0.1250, 0.0]) < 1e-4)
    lpc = vb.lpc(auto, 4)
    assert np.all(np.abs(lpc[0] - [1.0, -1.3122, 0.8660, -0.0875, -0.0103]) < 1e-4)
    r2, a2 = vb.autocorr_lpc(s[None, :], 4, normalize=True)
    assert np.all(np.abs(a2[0] - lpc[0]) < 1e-12)
    # lpc_mut (src/spectrum.rs:62-84) also leaves the reflection coefficients in `kc`: kc[i-1] is ac[i] as first set
    ac, kc = vb.lpc_mut(auto, 4)
    assert np.array_equal(ac, lpc) and kc.shape == (1, 4) and kc[0, 3] == ac[0, 4]
    a1, k1 = vb.lpc_mut(auto, 1)
    assert k1[0, 0] == a1[0, 1] == kc[0, 0]              # order-1 solution: a1 = k1 = -r1/r0
    assert abs(kc[0, 0] + auto[0, 1] / auto[0, 0]) < 1e-15


@pytest.mark.parametrize("n,p,norm", [(512, 12, False), (512, 12, True), (1200, 12, False), (256, 8, False),
                                      (700, 16, True), (512, 10, False), (300, 20, False), (1024, 46, False), (512, 62, True)])
def test_autocorr_lpc(vb, oracle, n, p, norm):
    rng = np.random.default_rng(n + p)
    t = np.arange(n)
    x = np.stack([np.sin(2 * np.pi * (0.01 + 0.002 * k) * t) * 0.5 + 0.2 * rng.standard_normal(n) for k in range(8)])
    x *= oracle.window("hanning", n)
    r, a = vb.autocorr_lpc(x, p, normalize=norm)
    for f in range(8):
        er = oracle.autocorrelate(x[f], p + 1)
        if norm:
            er = oracle.normalize(er)
        ea = oracle.lpc(er, p)
        assert np.all(rel_close(r[f], er)), (f, "r")
        assert np.all(rel_close(a[f], ea)), (f, "lpc", np.max(np.abs(a[f] - ea)))


# ---- Burg -------------------------------------------------------------------------------------

def test_lpc_praat_kat(vb, oracle):
    """src/spectrum.rs:512-525 test_lpc_praat (1e-10) through the GPU path."""
    src = np.array(list(range(1, 11)) + list(range(10, 0, -1)), dtype=np.float64)
    co, st = vb.lpc_praat(src[None, :], 5)
    exp = [-2.529731754197289, 2.6138925001574935, -1.6951059551991234, 0.7776548472652218, -0.15008712022777612]
    assert st[0] == 0 and np.all(np.abs(co[0] - exp) < 1e-10)


@pytest.mark.parametrize("n,p", [(512, 12), (1200, 12), (1024, 10), (100, 5), (2049, 13), (4096, 8), (30, 4), (513, 30),
                                 (64, 20), (128, 16), (200, 17), (2000, 30), (1025, 1), (40, 30),    # every lane-group shape
                                 (1024, 46), (600, 33), (2048, 62), (400, 62), (5000, 46)])           # orders above 30 (44.1 kHz material: sr / 1000 + 2)
def test_lpc_praat(vb, oracle, n, p):
    rng = np.random.default_rng(n * 31 + p)
    t = np.arange(n)
    x = np.stack([0.6 * np.sin(2 * np.pi * (0.013 + 0.004 * k) * t) + 0.3 * np.sin(2 * np.pi * 0.11 * t + k)
                  + 0.05 * rng.standard_normal(n) for k in range(6)])
    x[5] = 0.0                                        # all-zero frame -> Err(LPC) (src/spectrum.rs:123-125)
    co, st = vb.lpc_praat(x, p)
    for f in range(6):
        es, ec = oracle.lpc_burg(x[f], p)
        assert st[f] == es, f
        if es == 0:
            assert np.all(rel_close(co[f], ec)), (f, np.max(np.abs(co[f] - ec)))
        else:
            assert np.all(co[f] == 0.0)


# ---- polynomial ---------------------------------------------------------------------------------

def test_roots_kats(vb, oracle):
    """src/polynomial.rs:281-362 known answers through the GPU path (values AND discovery order)."""
    z = vb.laguerre(np.array([[1.0, 2.5, 2.0, 3.0]]), complex(-64.0, -64.0))[0]
    assert abs(z.real - -0.1070229535872) < 1e-8 and abs(z.imag - -0.8514680262155) < 1e-8
    r, st = vb.find_roots(np.array([[1.0, 2.5]]))
    assert st[0] == 0 and abs(r[0, 0] - (-0.4)) < 1e-12
    r, st = vb.find_roots(np.array([[1.0, 2.5, -2.0]]))
    assert st[0] == 0 and abs(r[0, 0] - (-0.31872930440884)) < 1e-12 and abs(r[0, 1] - 1.5687293044088) < 1e-12
    r, st = vb.find_roots(np.array([[1.0, -2.5, 2.0]]))
    assert abs(r[0, 0] - complex(0.625, -0.33071891388307)) < 1e-12
    assert abs(r[0, 1] - complex(0.625, 0.33071891388307)) < 1e-12
    r, st = vb.find_roots(np.array([[1.0, 2.5, -2.0, -3.0]]))
    for a, b in zip(r[0, :3], [-1.1409835232292, -0.35308705904629, 0.82740391560878]):
        assert abs(a - b) < 1e-6
    r, st = vb.find_roots(np.array([[1.0, 0.0, 0.0]]))
    assert st[0] == 2                                  # Err(Polynomial), src/polynomial.rs:95
    # src/spectrum.rs:615-633: the 4th root found must be the 4045 Hz one
    co = [-0.80098309, 1.20869679, -1.61846677, 0.86630291, -1.44203292, 0.93621726, -0.58772811, 0.65949051]
    r, st = vb.find_roots(np.array([([1.0] + co)[::-1]]))
    res, cnt = vb.to_resonance(r[:, 3:4], 11025.0)
    assert cnt[0] == 1 and abs(res[0, 0, 0] - 4045.196) < 1.0


def test_div_polynomial(vb, oracle):
    """Polynomial::div_polynomial_mut (src/polynomial.rs:155-195): quotient, remainder and the Err on a zero divisor
    against the oracle's literal restatement (the reference's own test is commented out, :216-267)."""
    rng = np.random.default_rng(21)
    P = (rng.uniform(-1, 1, (40, 8)) + 1j * rng.uniform(-1, 1, (40, 8)))
    P[5, 5:] = 0.0                                     # lower degree than the slice
    P[6, :] = 0.0; P[6, 0] = 2.0                       # degree 0
    O = rng.uniform(-2, 2, 40) + 1j * rng.uniform(-2, 2, 40)
    O[7] = 0.0                                         # "Tried to divide by zero"
    q, r, st = vb.div_polynomial(P, O)
    for f in range(P.shape[0]):
        es, eq, er = oracle.div_polynomial(P[f], complex(O[f]))
        assert st[f] == es, f
        assert np.allclose(q[f], eq, rtol=1e-12, atol=1e-13) and np.allclose(r[f], er, rtol=1e-12, atol=1e-13), f
    assert st[7] == 2
    # (x - 1)(x + 2) = x^2 + x - 2 divided by (x + 2): quotient x - 1, remainder 0
    q, r, st = vb.div_polynomial(np.array([[-2.0, 1.0, 1.0]]), np.array([2.0]))
    assert st[0] == 0 and np.allclose(q[0], [-1.0, 1.0, 0.0]) and np.allclose(r[0], 0.0)


def test_roots_f32_kats(vb, oracle):
    """src/polynomial.rs:336-386, the Complex<f32> instantiation through the GPU path."""
    r, st = vb.find_roots_f32(np.array([[1.0, -2.5, 2.0]]))
    assert st[0] == 0 and r.dtype == np.complex64
    exp = [np.complex64(complex(0.625, -0.33071891388307)), np.complex64(complex(0.625, 0.33071891388307))]
    for a, b in zip(r[0, :2], exp):
        assert abs(float(a.real) - float(b.real)) < 1e-12 and abs(float(a.imag) - float(b.imag)) < 1e-12
    r, st = vb.find_roots_f32(np.array([[1.0, 2.5, -2.0, -3.0]]))
    for a, b in zip(r[0, :3], [-1.1409835232292, -0.35308705904629, 0.82740391560878]):
        assert abs(float(a.real) - np.float32(b)) < 1e-6 and abs(float(a.imag)) < 1e-6
    lpc = [1.0, -0.99640256, 0.25383306, -0.25471634, 0.5084799, -0.0685858, -0.35042483, 0.07676613, -0.12874511,
           0.11829436, 0.023972526]
    z = vb.laguerre_f32(np.array([lpc]), complex(-64.0, -64.0))[0]
    assert np.isfinite(z.real) and np.isfinite(z.imag)
    zo = oracle.laguerre_f32(lpc, complex(-64.0, -64.0))
    assert abs(complex(z) - complex(zo)) <= 1e-4 * abs(complex(zo))
    r, st = vb.find_roots_f32(np.array([[1.0, 0.0, 0.0]]))
    assert st[0] == 2


def test_find_roots_f32_random(vb, oracle):
    """Random real-coefficient f32 polynomials against the f32 oracle: same root SET within single-precision
    conditioning (discovery order can differ when two Laguerre limits are rounding-close)."""
    rng = np.random.default_rng(12)
    P = rng.uniform(-1.0, 1.0, (64, 9)).astype(np.float32)
    P[:, -1] = 1.0
    r, st = vb.find_roots_f32(P.astype(np.complex64))
    for f in range(P.shape[0]):
        es, er = oracle.find_roots_f32(P[f].astype(np.complex64))
        assert st[f] == es
        if es != 0:
            continue
        g = np.sort_complex(r[f, :er.size].astype(np.complex128))
        e = np.sort_complex(er.astype(np.complex128))
        # residual check (conditioning-independent): every GPU root is a root of the polynomial to f32 accuracy
        pv = np.polyval(P[f, ::-1].astype(np.float64), r[f, :er.size].astype(np.complex128))
        scale = np.polyval(np.abs(P[f, ::-1]).astype(np.float64), np.abs(r[f, :er.size]).astype(np.float64))
        assert np.all(np.abs(pv) <= 2e-4 * scale), (f, np.abs(pv) / scale)
        assert g.size == e.size


def test_find_roots_random(vb, oracle):
    rng = np.random.default_rng(11)
    for deg in (3, 5, 8, 12, 13, 20):
        polys = rng.standard_normal((40, deg + 1)) + 0j
        polys[:, -1] = 1.0
        polys[3, -1] = 0.0                              # lower effective degree (degree() < len-1)
        polys[4, :] = 0.0; polys[4, 0] = 1.0            # zero-degree -> Err(Polynomial)
        got, st = vb.find_roots(polys)
        for f in range(polys.shape[0]):
            es, er = oracle.find_roots_mut(polys[f])
            assert st[f] == es, (deg, f, st[f], es)
            if es == 0:
                g = got[f].copy()
                tol = 1e-7 * np.maximum(1.0, np.abs(er))
                if not np.all(np.abs(g - er) <= tol):
                    # The closed-form quadratic tail (src/polynomial.rs:131-139) orders a conjugate pair by
                    # the SIGN of the rounding noise in sqrt(b^2-4ac)'s imaginary part; only that pair may swap.
                    nz = int(np.max(np.nonzero(er)[0])) if np.any(er != 0) else 0
                    g[[nz - 1, nz]] = g[[nz, nz - 1]]
                assert np.all(np.abs(g - er) <= tol), (deg, f)


def test_to_resonance(vb, oracle):
    """src/spectrum.rs:461-468 test_resonances + random rows."""
    res, cnt = vb.to_resonance(np.array([[complex(-0.5, 0.86602540378444), complex(-0.5, -0.86602540378444)]]), 300.0)
    assert cnt[0] == 1 and abs(res[0, 0, 0] - 100.0) < 1e-8 and abs(res[0, 0, 1]) < 1e-8
    rng = np.random.default_rng(5)
    roots = (rng.uniform(0.3, 1.3, (50, 12)) * np.exp(1j * rng.uniform(-np.pi, np.pi, (50, 12))))
    res, cnt = vb.to_resonance(roots, SR)
    for f in range(50):
        e = oracle.to_resonance(roots[f], SR)
        assert cnt[f] == e.shape[0]
        assert np.all(rel_close(res[f, :cnt[f]], e, 1e-10))
        assert np.all(res[f, cnt[f]:] == 0.0)


# ---- tracker -----------------------------------------------------------------------------------

def test_formant_extractor_kat(vb):
    """src/spectrum.rs:527-567 test_formant_extractor (exact) through the GPU scan."""
    fr = np.array([[100.0, 150.0, 200.0, 240.0, 300.0], [110.0, 180.0, 210.0, 230.0, 310.0],
                   [230.0, 270.0, 290.0, 350.0, 360.0]])
    res = np.stack([fr, np.ones_like(fr)], axis=-1)
    est = np.array([[140.0, 1.0], [230.0, 1.0], [320.0, 1.0]])
    out = vb.estimate_formants(res, est)
    assert out[:, :, 0].tolist() == [[150.0, 240.0, 300.0], [180.0, 230.0, 310.0], [230.0, 270.0, 290.0]]


def test_estimate_formants_random_segments(vb, oracle):
    rng = np.random.default_rng(17)
    F, R = 400, 32
    res = np.zeros((F, R, 2))
    for f in range(F):
        c = rng.integers(0, 7)
        fr = np.sort(rng.uniform(60, 5000, c))
        if c >= 2 and rng.random() < 0.3:
            fr[1] = fr[0]                                # duplicates
        res[f, :c, 0] = fr
        res[f, :c, 1] = rng.uniform(10, 400, c)
    est0 = np.array([[320.0, 1.0], [1440.0, 1.0], [2760.0, 1.0], [3200.0, 1.0]])
    seg = np.array([0, 50, 51, 200, 399])
    status = np.zeros(F, dtype=np.int32); status[[7, 120]] = 1
    got = vb.estimate_formants(res, est0, seg_start=seg, frame_status=status)
    est = None
    for f in range(F):
        if f in seg:
            est = est0.copy()
        if status[f] == 0:
            est = oracle.estimate_formants(est, res[f])
        assert np.array_equal(got[f], est), f


def test_estimate_formants_long_scan(vb, oracle):
    """A long scan must equal the
    sequential reference scan exactly, also where the state does not forget: rows with 0-2 resonances leave
    stale estimates alive for hundreds of frames (Q13)."""
    rng = np.random.default_rng(23)
    F, R = 6000, 32
    res = np.zeros((F, R, 2))
    for f in range(F):
        sparse = (f // 700) % 2 == 1                      # alternating stretches of rich and nearly empty rows
        c = int(rng.integers(0, 3)) if sparse else int(rng.integers(3, 7))
        fr = np.sort(rng.uniform(60, 5000, c))
        res[f, :c, 0] = fr
        res[f, :c, 1] = rng.uniform(10, 400, c)
    est0 = np.array([[320.0, 1.0], [1440.0, 1.0], [2760.0, 1.0], [3200.0, 1.0]])
    seg = np.array([0, 17, 40, 1000, 1031, 1032, 2500, 5999])
    status = np.zeros(F, dtype=np.int32)
    status[rng.integers(0, F, 60)] = 1
    got = vb.estimate_formants(res, est0, seg_start=seg, frame_status=status)
    est = None
    for f in range(F):
        if f in seg:
            est = est0.copy()
        if status[f] == 0:
            est = oracle.estimate_formants(est, res[f])
        assert np.array_equal(got[f], est), f
    one = vb.estimate_formants(res, est0)                 # one 6000-frame utterance
    est = est0.copy()
    for f in range(F):
        est = oracle.estimate_formants(est, res[f])
        assert np.array_equal(one[f], est), f


# ---- find_formants ---------------------------------------------------------------------------------

def test_find_formants_wav_fixture(vb, oracle, golden_dir):
    """tests/lib.rs:44-90: short_sample.wav, rectangle Windower 1024/512, p = 10."""
    samples, sr = _read_wav16(os.path.join(golden_dir, "short_sample.wav"))
    est0 = np.array([[f, 1.0] for f in (320.0, 1440.0, 2760.0, 3200.0)])
    out = vb.find_formants(samples, sr, 10, est0, frame_len=1024, stride=512)
    assert out["formants"].shape == (4, 4, 2) and np.all(out["status"] == 0)
    exp = [[1030.92, 2724.53, 3719.48, 3200.0], [1032.08, 2689.09, 3705.75, 3200.0],
           [1025.91, 2695.68, 2695.68, 3709.67], [1042.90, 2696.43, 3704.22, 3709.67]]
    assert np.all(np.abs(out["formants"][:, :, 0] - exp) < 0.01)
    est = est0.copy()
    for t in range(4):
        st, est, res, co = oracle.find_formants(samples[t * 512:t * 512 + 1024], sr, 10, est)
        assert np.all(rel_close(out["coeffs"][t], co))
        assert np.all(np.abs(out["res"][t] - res) <= 1e-4 * np.abs(res) + 1e-9)
        assert np.all(np.abs(out["formants"][t] - est) <= 1e-4 * np.abs(est))


@pytest.mark.parametrize("n,hop,p", [(N48, H48, 12), (512, 480, 12), (1024, 512, 10), (1024, 512, 16), (400, 160, 8)])
def test_find_formants_synthetic(vb, oracle, audio, pkg, n, hop, p):
    F = min(pkg.frame_count(audio.size, n, hop), 420)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    seg = np.array([0, 100, 250])
    d = vb.to_device(audio)
    out = vb.find_formants(d, SR, p, est0, seg_start=seg, frame_len=n, stride=hop, n_frames=F)
    d.free()
    est = None
    bad = 0
    for t in range(F):
        if t in seg:
            est = est0.copy()
        st, est, res, co = oracle.find_formants(audio[t * hop:t * hop + n], SR, p, est)
        assert out["status"][t] == st
        assert np.all(rel_close(out["coeffs"][t], co)), t
        assert out["count"][t] == int(np.sum(res[:, 0] != 0.0)), t
        ok = np.all(np.abs(out["res"][t] - res) <= 1e-4 * np.abs(res) + 1e-9) and \
            np.all(np.abs(out["formants"][t] - est) <= 1e-4 * np.abs(est))
        bad += 0 if ok else 1
    assert bad == 0, f"{bad} of {F} frames outside 1e-4"


# ---- sinc / Brent ------------------------------------------------------------------------------

def _lag_curve(oracle, x):
    n = x.size
    r = oracle.normalize(oracle.autocorrelate(x, n)) / oracle.window("hanning_lag", n)
    return np.concatenate([r, np.zeros(n)])


def test_interpolate_sinc_points(vb, oracle, audio):
    x = audio[1000:1000 + N48] * oracle.window("hanning", N48)
    y = _lag_curve(oracle, x)
    b = N48 // 2
    offset, nx = -b - 1, 2 * b + 1
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(b + 2, 2 * b, 300), [b + 1 + 100.0, b + 1 + 100.0 + 5e-11, -1.0, nx + 5.0, b + 1.5]])
    for depth in (30, 1200):
        got, st = vb.interpolate_sinc(y, offset, nx, xs, depth)
        for i, xv in enumerate(xs):
            es, ev = oracle.interpolate_sinc(y, offset, nx, xv, depth)
            assert st[i] == es
            assert abs(got[i] - ev) <= 1e-9 * max(1.0, abs(ev)), (depth, xv, got[i], ev)


def test_improve_extremum_points(vb, oracle, audio):
    x = audio[20000:20000 + N48] * oracle.window("hanning", N48)
    y = _lag_curve(oracle, x)
    b = N48 // 2
    offset, nx = -b - 1, 2 * b + 1
    peaks = [k for k in range(1, b - 1) if y[k - 1] < y[k] > y[k + 1] and 80 < k < 590]
    ix = np.array([k + 0.01 * ((k * 7) % 10) - offset for k in peaks] + [0.0, float(nx), nx + 3.0])
    got, st = vb.improve_extremum(y, offset, nx, ix, 1200)
    for i, v in enumerate(ix):
        es, ex, ey = oracle.improve_extremum_sinc(y, offset, nx, v, 1200)
        assert st[i] == es
        assert abs(got[i, 0] - ex) <= 1e-6 * max(1.0, abs(ex)), (v, got[i], ex, ey)
        assert abs(got[i, 1] - ey) <= 1e-6 * max(1.0, abs(ey)), (v, got[i], ex, ey)


def test_improve_extremum_every_arm(vb, oracle, audio):
    """`Interpolation::{None, Parabolic, Sinc}` and `is_max` (src/periodic.rs:89-93,192-229): only Sinc with is_max = true
    is on the pitch path, the other arms are public surface of the crate."""
    x = audio[20000:20000 + N48] * oracle.window("hanning", N48)
    y = _lag_curve(oracle, x)
    b = N48 // 2
    offset, nx = -b - 1, 2 * b + 1
    peaks = [k for k in range(1, b - 1) if y[k - 1] < y[k] > y[k + 1] and 80 < k < 590]
    ix = np.array([k + 0.01 * ((k * 7) % 10) - offset for k in peaks[:12]] + [0.0, float(nx), nx + 3.0, 0.5, 1.25, 2.0])
    for interp, depth, is_max in ((0, 0, True), (1, 0, True), (1, 0, False), (2, 1200, False), (2, 30, True)):
        got, st = vb.improve_extremum_ex(y, offset, nx, ix, interp, depth, is_max)
        for i, v in enumerate(ix):
            es, ex, ey = oracle.improve_extremum(y, offset, nx, v, interp, depth, is_max)
            assert st[i] == es, (interp, is_max, v, st[i], es)
            if es == 0:
                tol = 1e-6 if interp == 2 else 1e-14
                assert abs(got[i, 0] - ex) <= tol * max(1.0, abs(ex)), (interp, is_max, v, got[i], ex, ey)
                assert abs(got[i, 1] - ey) <= tol * max(1.0, abs(ey)), (interp, is_max, v, got[i], ex, ey)
    # the parabolic arm indexes y[floor(ixmid) - 1 .. + 1] of the slice it is given: out of bounds is the reference's panic
    got, st = vb.improve_extremum_ex(y[:64], 0, 200, np.array([0.25, 62.5, 63.5, 100.0]), 1)
    exp = [oracle.improve_extremum(y[:64], 0, 200, v, 1)[0] for v in (0.25, 62.5, 63.5, 100.0)]
    assert list(st) == exp and exp[0] == 4 and exp[2] == 4


# ---- pitch ----------------------------------------------------------------------------------------

def test_pitch_kat(vb, oracle, pkg):
    """src/periodic.rs:484-499 test_pitch / examples/pitch_detection.rs: 150 Hz sine @44.1 kHz."""
    sig = oracle.sine(2049, 44100.0, 150.0)
    d = vb.to_device(sig)
    w = vb.window(pkg.WINDOW_HANNING, 2048)
    cand, cnt, st = vb.pitch(d, 44100.0, 0.2, 100.0, 500.0, kmax=4, frame_len=2048, stride=1024, n_frames=1, window=w)
    assert st[0] == 0 and cnt[0] == 2
    assert abs(cand[0, 0, 0] - 150.0) < 1e-2
    assert abs(cand[0, 0, 0] - 149.9999843470686) < 1e-4 * 150 and abs(cand[0, 0, 1] - 0.9997482091589159) < 1e-6
    assert cand[0, 1, 0] == 0.0 and cand[0, 1, 1] == 0.2
    # Q8 quantisation: 137.3 Hz @ 48 kHz -> 48000/350
    s2 = oracle.sine(1200, 48000.0, 137.3)
    cand, cnt, st = vb.pitch(s2[None, :] * oracle.window("hanning", 1200), 48000.0, 0.2, 75.0, 600.0)
    assert abs(cand[0, 0, 0] - 137.1428566729394) < 1e-4 * 137 and abs(cand[0, 0, 1] - 0.9985693856763005) < 1e-6


PITCH_STATS = []      # one dict per _check_pitch call; printed by test_pitch_parity_report (run last in this file)


def _check_pitch(vb, oracle, frames_windowed, sr, thr, fmin, fmax, kmax, stats=None, label=""):
    """Parity metric for Pitched::pitch.  Status and candidate COUNT are exact.  The Brent refinement
    stops at |dx| ~ 3e-5 lags and is chaotic below that (DESIGN.md "Brent sensitivity"), so values are
    compared within BASELINE's tolerance: every frequency within 1e-4 relative, strengths within 1e-4.
    Everything that is NOT an exact agreement is COUNTED and bounded:
      n_flip      candidates whose refinement ended on the other side of the integer-lag discontinuity (same
                  frequency, strength off by more than 1e-4)
      n_top_swap  frames whose top candidate (the PitchExtractor output) differs because the oracle's two best
                  strengths are closer than 1e-3 AND the GPU's top is the oracle's runner-up
      n_vuv_flip  of those, swaps between a voiced candidate and the unvoiced one (a voiced/unvoiced decision
                  change): allowed only inside a 1e-4 tie, i.e. inside the tolerance itself
      n_top_bad   any other top-candidate disagreement: zero.
    Bounds = what is observed plus a stated margin.  Observed on the whole corpus of this file (rounds 2 and 3: ~700 frames,
    ~2,400 fully compared candidates; gpurun_out/pitch_parity_stats.json): n_flip = n_top_swap = 0.  Margin: one event per
    call (a single candidate whose chaotic tail lands elsewhere after a change of summation order must not turn the suite
    red), and test_zz_pitch_parity_report bounds the TOTALS at 0.2 % of the frames / 0.1 % of the candidates -- a fifth
    and a tenth of the round-2 bounds."""
    cand, cnt, st = vb.pitch(frames_windowed, sr, thr, fmin, fmax, kmax=kmax)
    F = frames_windowed.shape[0]
    n_cand = n_flip = n_top_bad = n_top_swap = n_vuv_flip = n_vuv_outside = 0
    worst_gap = 0.0
    for f in range(F):
        es, ec, en = oracle.pitch(frames_windowed[f], sr, thr, fmin, fmax)
        assert st[f] == es, (f, st[f], es)
        assert cnt[f] == (en if es == 0 else 0), (f, cnt[f], en)
        k = min(kmax, en)
        assert np.all(cand[f, k:] == 0.0)
        if es != 0:
            assert np.all(cand[f] == 0.0)
            continue
        top_ok = abs(cand[f, 0, 0] - ec[0, 0]) <= 1e-4 * abs(ec[0, 0]) and abs(cand[f, 0, 1] - ec[0, 1]) <= 1e-4
        if not top_ok:
            gap = abs(ec[0, 1] - ec[1, 1]) if en > 1 else np.inf
            is_runner_up = en > 1 and abs(cand[f, 0, 0] - ec[1, 0]) <= 1e-4 * abs(ec[1, 0]) and abs(cand[f, 0, 1] - ec[1, 1]) <= 1e-3
            if gap < 1e-3 and is_runner_up:
                n_top_swap += 1
                worst_gap = max(worst_gap, gap)
                if (cand[f, 0, 0] == 0.0) != (ec[0, 0] == 0.0):
                    n_vuv_flip += 1
                    n_vuv_outside += int(gap > 1e-4)
            else:
                n_top_bad += 1
        if kmax >= en:
            # full list: compare as sets ordered by frequency (strength order may permute within tolerance)
            g = cand[f, :k][np.argsort(cand[f, :k, 0], kind="stable")]
            e = ec[:k][np.argsort(ec[:k, 0], kind="stable")]
            assert np.all(np.abs(g[:, 0] - e[:, 0]) <= 1e-4 * np.abs(e[:, 0])), (f, "frequency")
            ds = np.abs(g[:, 1] - e[:, 1])
            n_cand += k
            n_flip += int(np.sum(ds > 1e-4))
            # the GPU list itself must be sorted by descending strength
            assert np.all(np.diff(cand[f, :k, 1]) <= 0.0), (f, "order")
    rec = dict(label=label, frames=F, kmax=kmax, n_cand=n_cand, n_flip=n_flip, n_top_bad=n_top_bad,
               n_top_swap=n_top_swap, n_vuv_flip=n_vuv_flip, n_vuv_outside_tol=n_vuv_outside, worst_tie_gap=worst_gap)
    PITCH_STATS.append(rec)
    if stats is not None:
        stats.update(rec)
    assert n_top_bad == 0, rec
    assert n_vuv_outside == 0, rec
    assert n_top_swap <= 1, rec
    assert n_flip <= max(1, n_cand // 1000), rec
    return 0


def test_pitch_synthetic_voiced_and_unvoiced(vb, oracle, audio, pkg):
    F = pkg.frame_count(audio.size, N48, H48)
    idx = list(range(0, F, 9))                          # spans the voiced glide and the noise-only second
    x = _frames(audio, N48, H48, idx) * oracle.window("hanning", N48)
    assert _check_pitch(vb, oracle, x, SR, 0.2, 75.0, 600.0, 8, label="synthetic voiced + unvoiced") == 0


def test_pitch_full_candidate_list(vb, oracle, audio):
    x = _frames(audio, N48, H48, [5, 150, 260, 300]) * oracle.window("hanning", N48)   # 260/300: unvoiced
    assert _check_pitch(vb, oracle, x, SR, 0.2, 75.0, 600.0, 64) == 0


def test_pitch_whole_vec(vb, oracle, audio, pkg):
    """src/periodic.rs:452-454 returns EVERY candidate.  kmax = VBX_PITCH_MAX_CANDIDATES(frame_len) (the LDS-resident
    list, nothing pruned) must hold the oracle's whole sorted Vec -- unvoiced frames carry ~175 candidates, more than
    the 64 entries of the lane-resident list -- and the two-call count/fill protocol must return the same rows."""
    F = pkg.frame_count(audio.size, N48, H48)
    idx = list(range(3, F, 23)) + [260, 300, 420, 455]                       # voiced glide + noise-only seconds
    x = _frames(audio, N48, H48, idx) * oracle.window("hanning", N48)
    kfull = pkg.pitch_max_candidates(N48)
    assert _check_pitch(vb, oracle, x, SR, 0.2, 75.0, 600.0, kfull, label="whole Vec, N=1200") == 0
    full, cnt, st = vb.pitch(x, SR, 0.2, 75.0, 600.0, kmax=kfull)
    assert cnt.max() > 64 and cnt.max() <= kfull
    # two-call protocol: count with kmax = 1, then kmax = max(count)
    _, cnt1, st1 = vb.pitch(x, SR, 0.2, 75.0, 600.0, kmax=1)
    assert np.array_equal(cnt1, cnt) and np.array_equal(st1, st)
    again, cnt2, _ = vb.pitch(x, SR, 0.2, 75.0, 600.0, kmax=int(cnt1.max()))
    assert np.array_equal(again, full[:, :int(cnt1.max())]) and np.array_equal(cnt2, cnt)
    # the lane-resident lists (4 <= kmax <= 64, pruned) are bit for bit the head of the whole Vec; kmax = 1 (the other
    # class of kmax, see test_pitch_topk_is_the_prefix_of_the_full_list) within the Brent iteration's scatter
    for kmax in (4, 8, 64):
        head, c, s_ = vb.pitch(x, SR, 0.2, 75.0, 600.0, kmax=kmax)
        assert np.array_equal(head, full[:, :kmax]) and np.array_equal(c, cnt) and np.array_equal(s_, st), kmax
    head, c, s_ = vb.pitch(x, SR, 0.2, 75.0, 600.0, kmax=1)
    assert np.array_equal(c, cnt) and np.array_equal(s_, st)
    assert np.all(np.abs(head[:, 0, 0] - full[:, 0, 0]) <= 1e-6 * np.abs(full[:, 0, 0])) and np.all(np.abs(head[:, 0, 1] - full[:, 0, 1]) <= 1e-5)
    # a kmax between 64 and the count: still the head of the same list
    mid, _, _ = vb.pitch(x, SR, 0.2, 75.0, 600.0, kmax=100)
    assert np.array_equal(mid, full[:, :100])


@pytest.mark.parametrize("n", [256, 513, 2048, 4096])
def test_pitch_whole_vec_other_lengths(vb, oracle, audio, pkg, n):
    """The whole Vec through the direct-sum (matrix-core) kernel, which serves every frame length but 1200."""
    x = _frames(audio, n, 211, range(0, 40, 5 if n <= 2048 else 13)) * oracle.window("hanning", n)
    x[-1] = np.random.default_rng(n).standard_normal(n) * oracle.window("hanning", n)      # many candidates
    kfull = pkg.pitch_max_candidates(n)
    assert _check_pitch(vb, oracle, x, SR, 0.45, 60.0, 20000.0, kfull, label=f"whole Vec, N={n}") == 0
    full, cnt, st = vb.pitch(x, SR, 0.45, 60.0, 20000.0, kmax=kfull)
    head, c, s_ = vb.pitch(x, SR, 0.45, 60.0, 20000.0, kmax=5)
    assert np.array_equal(head, full[:, :5]) and np.array_equal(c, cnt) and np.array_equal(s_, st)


@pytest.mark.parametrize("thr", [0.0, 0.2, 0.6, 0.999, 5.0])
def test_pitch_topk_is_the_prefix_of_the_full_list(vb, oracle, audio, pkg, thr):
    """The refine kernel skips candidates that provably cannot reach the first kmax entries (DESIGN.md
    "exact top-k pruning").  What is returned must be the prefix of the unpruned list (a list that never fills,
    kmax >= count, disables the bound), and count/status must not change.  BIT FOR BIT inside each of the two classes of
    kmax -- {1, 2, 3} and {4, ...}: from kmax = 4 on, frames with few candidates refine them four at a time with 16 lanes each
    instead of one at a time with 64, which changes the last bits of a candidate's sinc sums and with them where the
    chaotic tail of its Brent iteration lands.  ACROSS the classes: the same candidates in the same order within the Brent
    iteration's own scatter (1e-6 relative in Hz, 1e-5 in strength: a hundred times inside the parity tolerance), except where
    two neighbours in the list are closer than that in strength (they may swap)."""
    F = pkg.frame_count(audio.size, N48, H48)
    x = _frames(audio, N48, H48, list(range(0, F, 7))) * oracle.window("hanning", N48)
    full, cnt_full, st_full = vb.pitch(x, SR, thr, 75.0, 600.0, kmax=64)
    assert cnt_full.min() < 64 < cnt_full.max()       # both unpruned and pruned frames in the 64-deep run
    for kmax in (4, 8, 15, 16, 33):
        cand, cnt, st = vb.pitch(x, SR, thr, 75.0, 600.0, kmax=kmax)
        assert np.array_equal(cnt, cnt_full) and np.array_equal(st, st_full)
        assert np.array_equal(cand, full[:, :kmax]), kmax
    c3, cnt, st = vb.pitch(x, SR, thr, 75.0, 600.0, kmax=3)
    assert np.array_equal(cnt, cnt_full) and np.array_equal(st, st_full)
    for kmax in (1, 2):
        cand, cnt, st = vb.pitch(x, SR, thr, 75.0, 600.0, kmax=kmax)
        assert np.array_equal(cnt, cnt_full) and np.array_equal(st, st_full)
        assert np.array_equal(cand, c3[:, :kmax]), kmax
    # across the classes
    a, b = c3, full[:, :3]
    close = (np.abs(a[:, :, 0] - b[:, :, 0]) <= 1e-6 * np.abs(b[:, :, 0])) & (np.abs(a[:, :, 1] - b[:, :, 1]) <= 1e-5)
    rows = np.nonzero(~np.all(close, axis=1))[0]
    for f in rows:                                    # a swap of two entries whose strengths are closer than the scatter
        sa, sb = np.sort(a[f, :, 1]), np.sort(b[f, :, 1])
        assert np.all(np.abs(sa - sb) <= 1e-5) and np.min(np.abs(np.diff(full[f, :4, 1]))) <= 2e-5, (f, a[f], b[f])
    assert rows.size <= max(1, x.shape[0] // 100), rows.size
    # and the prefix agrees with the oracle's sorted Vec
    assert _check_pitch(vb, oracle, x[::5], SR, thr, 75.0, 600.0, 1) == 0
    assert _check_pitch(vb, oracle, x[::9], SR, thr, 75.0, 600.0, 8) == 0


def test_roctx_ranges_do_not_change_results(pkg, oracle, audio, monkeypatch):
    """VBX_ROCTX=1: every kernel group is also a roctx range (rocprofv3 --marker-trace shows which entry point a kernel belongs
    to).  Without a profiler attached the ranges go nowhere; the outputs are the same bits."""
    x = _frames(audio, N48, H48, list(range(0, 400, 7))) * oracle.window("hanning", N48)
    res = []
    for on in (False, True):
        if on:
            monkeypatch.setenv("VBX_ROCTX", "1")
        v = pkg.VoxBox(0)
        if on:
            monkeypatch.delenv("VBX_ROCTX")
        try:
            res.append((v.pitch(x, SR, 0.2, 75.0, 600.0, kmax=4), v.mfcc(x, 13, (100.0, 8000.0), SR)))
        finally:
            v.close()
    (pa, ma), (pb, mb) = res
    assert all(np.array_equal(a, b) for a, b in zip(pa, pb)) and all(np.array_equal(a, b) for a, b in zip(ma, mb))


def test_pitch_work_counters(vb, oracle, audio, pkg):
    """vbx_profile_pitch_work reports the work the refine kernel executed (bench.py's FP64 roofline uses it)."""
    F = pkg.frame_count(audio.size, N48, H48)
    x = _frames(audio, N48, H48, list(range(0, F, 11))) * oracle.window("hanning", N48)
    work = {}
    vb.profile(True)
    try:
        for kmax in (1, 64):
            vb.profile_reset()
            _, cnt, st = vb.pitch(x, SR, 0.2, 75.0, 600.0, kmax=kmax)
            work[kmax] = vb.profile_pitch_work()
            assert work[kmax][0] == x.shape[0] and work[kmax][1] == int(np.sum(cnt[st == 0] - 1))
    finally:
        vb.profile(False)
    assert 0 < work[1][2] < work[64][2] and 0 < work[1][3] < work[64][3]     # pruning skips evaluations
    assert work[64][2] >= work[64][1]                                  # unpruned: every candidate is evaluated


# 1280 / 1281: one / two autocorrelation passes of the direct kernel (VBX_PITCH_MFMA=1); 512..4096 take the FFT kernels
@pytest.mark.parametrize("n", [64, 100, 256, 511, 512, 513, 703, 704, 800, 1023, 1024, 1025, 1103, 1199, 1201, 1280, 1281, 1600, 2047, 2048,
                               2049, 4096])
def test_pitch_other_frame_lengths(vb, oracle, audio, n):
    x = _frames(audio, n, 211, range(0, 40, 3 if n <= 2048 else 8)) * oracle.window("hanning", n)
    assert _check_pitch(vb, oracle, x, SR, 0.45, 60.0, 2000.0, 8) == 0


@pytest.mark.parametrize("n,rect", [(513, False), (1103, False), (1103, True), (1199, False), (1601, False), (2047, True), (2049, False), (4095, False)])
def test_pitch_odd_frame_lengths_read_the_last_lag(vb, oracle, audio, n, rect):
    """An odd frame length makes improve_extremum's `ixmid >= nx` arm (src/periodic.rs:194) return y[n - 1], the LAST lag of
    the curve, where the lag window is ~1e-17: candidates at the edge of the search range take their strength from it.
    Unvoiced frames have such candidates (found by tools/soak_parity.py at n = 1103: the FFT kernels' rounding error,
    divided by that window value, used to win the frame).  The FFT kernels compute the last lags as the reference does."""
    hop = 211
    F = (audio.size - n) // hop + 1
    lo, hi = int(2.0 * 48000 / hop), int(3.0 * 48000 / hop)                   # the unvoiced second (4..5 s of the stream)
    idx = list(range(lo, min(hi, F), 2)) + list(range(0, lo, 40))
    w = np.ones(n) if rect else oracle.window("hanning", n)
    x = _frames(audio, n, hop, idx) * w
    assert _check_pitch(vb, oracle, x, SR, 0.2, 75.0, 600.0, 2, label=f"odd n = {n}") == 0


@pytest.mark.parametrize("n", [4, 5, 6, 8, 13, 16, 31])
def test_pitch_tiny_frames(vb, oracle, audio, n):
    """Frames of a few samples: the depth clips, the x > nx / x < 0 branches and the panics of the reference's
    index arithmetic all live here; status, count and values must follow the oracle."""
    x = _frames(audio, n, 7, range(0, 200, 5))
    cand, cnt, st = vb.pitch(x, SR, 0.1, 1000.0, 30000.0, kmax=8)
    for f in range(x.shape[0]):
        es, ec, en = oracle.pitch(x[f], SR, 0.1, 1000.0, 30000.0)
        assert st[f] == es and cnt[f] == (en if es == 0 else 0), (n, f, st[f], es, cnt[f], en)
        if es == 0:
            k = min(8, en)
            assert np.all(np.abs(cand[f, :k, 0] - ec[:k, 0]) <= 1e-4 * np.abs(ec[:k, 0]) + 1e-12), (n, f)
            assert np.all(np.abs(cand[f, :k, 1] - ec[:k, 1]) <= 1e-4), (n, f)


def test_pitch_odd_signals(vb, oracle):
    """Noise, pure and clipped tones, square waves, chirps, impulses, DC, 1e-150 and 1e120 amplitudes: status and
    candidate count exact, top candidate within BASELINE tolerance, and the kmax = 1, 3 outputs bit-identical to
    the head of the kmax = 64 list."""
    N, rng = 1200, np.random.default_rng(123)
    t = np.arange(N) / SR
    w = oracle.window("hanning", N)
    gens = [lambda: rng.standard_normal(N),
            lambda: np.sin(2 * np.pi * rng.uniform(60, 700) * t + rng.uniform(0, 6)),
            lambda: np.sign(np.sin(2 * np.pi * rng.uniform(80, 400) * t)),
            lambda: np.sin(2 * np.pi * (100 + 3000 * t) * t),
            lambda: np.bincount(rng.integers(0, N, 5), minlength=N).astype(np.float64),
            lambda: 0.5 + 0.01 * rng.standard_normal(N),
            lambda: 1e-150 * np.sin(2 * np.pi * 200 * t),
            lambda: 1e120 * np.sin(2 * np.pi * 150 * t),
            lambda: np.sin(2 * np.pi * 120 * t) * (1 + 0.5 * np.sin(2 * np.pi * 7 * t)) + 0.2 * rng.standard_normal(N),
            lambda: np.clip(3 * np.sin(2 * np.pi * rng.uniform(75, 600) * t), -1, 1)]
    X = np.array([gens[i % len(gens)]() * w for i in range(200)])
    res = {k: vb.pitch(X, SR, 0.2, 75.0, 600.0, kmax=k) for k in (1, 3, 8, 64)}
    for k in (1, 3, 8):
        assert np.array_equal(res[k][1], res[64][1]) and np.array_equal(res[k][2], res[64][2])
    # bit for bit inside a class of kmax ({1, 2, 3} and {4, ...}: test_pitch_topk_is_the_prefix_of_the_full_list)
    assert np.array_equal(res[1][0], res[3][0][:, :1]) and np.array_equal(res[8][0], res[64][0][:, :8])
    # (the head of the kmax = 64 list is held to the oracle below, through _check_pitch at kmax = 8)
    assert _check_pitch(vb, oracle, X[::4], SR, 0.2, 75.0, 600.0, 8, label="odd signals, kmax 8") == 0
    # the head of the list against the oracle, every disagreement counted (see _check_pitch)
    assert _check_pitch(vb, oracle, X, SR, 0.2, 75.0, 600.0, 1, label="odd signals") == 0


def test_pitch_nonfinite_input(vb, oracle, audio):
    """NaN / inf samples: the reference's sort panics on NaN strengths (status 3) -- pruning must not hide that."""
    x = _frames(audio, N48, H48, [5, 40, 150, 260]) * oracle.window("hanning", N48)
    x[1, 100] = np.nan
    x[2, 7] = np.inf
    x[3, :] = 0.0
    for kmax in (1, 8):
        cand, cnt, st = vb.pitch(x, SR, 0.2, 75.0, 600.0, kmax=kmax)
        for f in range(x.shape[0]):
            es, ec, en = oracle.pitch(x[f], SR, 0.2, 75.0, 600.0)
            assert st[f] == es and cnt[f] == (en if es == 0 else 0), (kmax, f, st[f], es, cnt[f], en)
    cand, cnt, st = vb.pitch(x[:1], SR, float("nan"), 75.0, 600.0, kmax=1)       # NaN threshold: unwrap() panics
    es, ec, en = oracle.pitch(x[0], SR, float("nan"), 75.0, 600.0)
    assert st[0] == es


def test_pitch_edge_frames(vb, oracle):
    z = np.zeros((2, 512))
    z[1, :] = 1e-3                                        # constant frame
    cand, cnt, st = vb.pitch(z, SR, 0.2, 75.0, 600.0, kmax=4)
    for f in range(2):
        es, ec, en = oracle.pitch(z[f], SR, 0.2, 75.0, 600.0)
        assert st[f] == es and cnt[f] == en
        if es == 0:
            assert np.all(np.abs(cand[f, :min(4, en)] - ec[:4]) <= 1e-9)


# ---- MFCC --------------------------------------------------------------------------------------------

def test_dct_kat(vb):
    """src/spectrum.rs:604-613 test_dct."""
    d = vb.dct(np.array([[0.2, 0.3, 0.4, 0.3]]))
    assert np.all(np.abs(d[0] - [2.4, -0.26131, -0.28284, 0.10823]) < 1e-5)


def test_mfcc_not_nan(vb):
    """src/spectrum.rs:592-602 test_mfcc_not_nan."""
    m, st = vb.mfcc(np.zeros((1, 512)), 13, (100.0, 8000.0), 22050.0)
    assert st[0] == 0 and np.all(np.isfinite(m)) and abs(m[0, 0] - 2.6e-9) < 1e-15


@pytest.mark.parametrize("n,k,lo,hi,sr", [(1200, 13, 100.0, 8000.0, 48000.0), (256, 26, 133.0, 6855.0, 22050.0),
                                          (512, 13, 100.0, 8000.0, 22050.0), (1024, 20, 0.0, 4000.0, 16000.0),
                                          (509, 13, 100.0, 8000.0, 22050.0),      # prime length: Goertzel kernel
                                          (1155, 13, 50.0, 11000.0, 22050.0),     # odd factors (3*5*7*11)
                                          (1000, 40, 0.0, 8000.0, 16000.0),       # every bin up to n/2 needed
                                          (2400, 13, 100.0, 8000.0, 96000.0), (94, 5, 300.0, 3000.0, 8000.0),
                                          (2048, 13, 100.0, 8000.0, 48000.0), (2048, 40, 0.0, 24000.0, 48000.0),   # FFT kernels: every bin
                                          (4096, 26, 50.0, 16000.0, 44100.0), (1024, 64, 0.0, 11025.0, 22050.0),
                                          (1200, 40, 0.0, 24000.0, 48000.0)])
def test_mfcc(vb, oracle, audio, n, k, lo, hi, sr):
    x = _frames(audio, n, 977, range(0, 60, 4)) * oracle.window("hanning", n) * 40.0
    m, st = vb.mfcc(x, k, (lo, hi), sr)
    for f in range(x.shape[0]):
        es, em = oracle.mfcc(x[f], k, lo, hi, sr)
        assert st[f] == es
        assert np.all(rel_close(m[f], em, 1e-6)), (f, np.max(np.abs(m[f] - em)))


def test_mfcc_four_kernels_agree(pkg, oracle, audio, monkeypatch):
    """N = 1200 takes the FFT kernel (the forward half of the fused spectral kernel: 1024, 1200, 2048, 4096); VBX_MFCC_MFMA=1
    keeps the matrix-core two-stage DFT (other composite lengths), VBX_MFCC_DFT2=1 forces the vector two-stage kernel (lengths
    whose factorisation does not fit the MFMA tiles) and VBX_MFCC_GOERTZEL=1 the Goertzel kernel (prime lengths).  The four
    must agree with each other far inside the oracle tolerance, and differ in the last bits (four kernels ran)."""
    x = _frames(audio, N48, 977, range(0, 60, 4)) * oracle.window("hanning", N48) * 40.0
    res = {}
    for name, var in (("fft", None), ("mfma", "VBX_MFCC_MFMA"), ("dft2", "VBX_MFCC_DFT2"), ("goertzel", "VBX_MFCC_GOERTZEL")):
        if var:
            monkeypatch.setenv(var, "1")
        v = pkg.VoxBox(0)
        if var:
            monkeypatch.delenv(var)
        try:
            res[name] = v.mfcc(x, 13, (100.0, 8000.0), SR)
        finally:
            v.close()
    for a in ("fft", "dft2", "goertzel"):
        assert np.array_equal(res["mfma"][1], res[a][1]) and not np.array_equal(res["mfma"][0], res[a][0])
        assert np.all(rel_close(res["mfma"][0], res[a][0], 1e-9)), a
    assert not np.array_equal(res["dft2"][0], res["goertzel"][0]) and not np.array_equal(res["fft"][0], res["dft2"][0])


@pytest.mark.parametrize("n,k,lo,hi,sr", [(1103, 13, 100.0, 8000.0, 44100.0), (3000, 13, 100.0, 8000.0, 48000.0), (2500, 13, 100.0, 8000.0, 48000.0),
                                          (997, 20, 50.0, 6000.0, 22050.0), (2049, 26, 133.0, 6855.0, 22050.0), (3301, 13, 100.0, 8000.0, 48000.0),
                                          (601, 13, 100.0, 8000.0, 48000.0), (1200, 13, 100.0, 8000.0, 48000.0), (700, 40, 0.0, 10000.0, 48000.0),
                                          (64, 5, 300.0, 3000.0, 8000.0), (1601, 64, 0.0, 8000.0, 16000.0),
                                          (4000, 13, 100.0, 8000.0, 48000.0), (3601, 13, 100.0, 8000.0, 44100.0), (4095, 20, 0.0, 11000.0, 48000.0)])
def test_mfcc_chirp_z_kernel(pkg, oracle, audio, monkeypatch, n, k, lo, hi, sr):
    """Frame lengths with no transform and no matrix-core factorisation of their own (1103 = 25 ms at 44.1 kHz is prime; 2500 and
    3000 need more bins than the two-stage kernel's tiles hold) took the chirp-z kernel (k_mfcc_czt.hip) by default until round 5;
    VBX_MFCC_CZT=1 sends every length through it that fits (n + top - 1 <= 4096), VBX_MFCC_CZT=0 none: both against the oracle,
    and -- where the default is the chirp-z kernel -- the two must differ in the last bits (two kernels ran)."""
    x = _frames(audio, n, 977, range(0, 40, 4)) * oracle.window("hanning", n) * 40.0
    res = {}
    interpolated = False
    for mode in ("1", "0", None):
        if mode is not None:
            monkeypatch.setenv("VBX_MFCC_CZT", mode)
        v = pkg.VoxBox(0)
        if mode is not None:
            monkeypatch.delenv("VBX_MFCC_CZT")
        try:
            res[mode] = v.mfcc(x, k, (lo, hi), sr)
            if mode is None:
                interpolated = bool(v.L.vbx_internal_last_mfcc_interp(v.ctx))
        finally:
            v.close()
    for f in range(x.shape[0]):
        es, em = oracle.mfcc(x[f], k, lo, hi, sr)
        for mode in res:
            assert res[mode][1][f] == es
            assert np.all(rel_close(res[mode][0][f], em, 1e-6)), (mode, f, np.max(np.abs(res[mode][0][f] - em)))
    if n != 1200:                                              # (a frame that IS a transform's keeps the FFT kernel either way)
        assert not np.array_equal(res["1"][0], res["0"][0])
    if n in (1103, 3000, 2500, 997, 2049, 3301, 601, 4000, 3601, 4095):   # (the last three: too long for one transform, split in two)
        # the default WAS the chirp-z kernel here (rounds 3-4); since round 5 it is the fused kernels' forward transform with
        # interpolated bins wherever that form exists (bins below a quarter of the transform, the staging inside the LDS budget:
        # not 997 / 2049 / 3601 / 4095 with these filters), within 1e-9 of the chirp-z kernel's exact arithmetic
        assert interpolated == (n in (1103, 3000, 2500, 3301, 601, 4000)), (n, interpolated)
        if interpolated:
            assert not np.array_equal(res[None][0], res["1"][0]) and np.all(rel_close(res[None][0], res["1"][0], 1e-10))
        else:
            assert np.array_equal(res[None][0], res["1"][0])
    if n in (1200, 64):
        assert np.array_equal(res[None][0], res["0"][0])       # ... and is not here (its own transform; too short)


@pytest.mark.parametrize("n,k,lo,hi,sr", [(1103, 13, 100.0, 8000.0, 44100.0), (997, 20, 50.0, 6000.0, 22050.0), (2500, 13, 100.0, 8000.0, 48000.0),
                                          (64, 5, 300.0, 3000.0, 8000.0), (1601, 64, 0.0, 8000.0, 16000.0), (4000, 13, 100.0, 8000.0, 48000.0)])
def test_mfcc_chirp_z_split_in_two_blocks(pkg, oracle, audio, monkeypatch, n, k, lo, hi, sr):
    """A frame too long for one transform (frame_len + top_bin - 1 > 4096) goes through the chirp-z kernel in two halves, each
    convolved with its own segment of the chirp, the complex results summed before the magnitude (vbx_mfcc_czt.hpp).
    VBX_MFCC_CZT_SPLIT=1 takes that form wherever it fits -- odd lengths (halves of unequal length), short ones, many
    coefficients: against the oracle at the MFCC tolerance, and against the one-block form within 1e-9 (same bins, another
    summation)."""
    x = _frames(audio, n, 977, range(0, 40, 4)) * oracle.window("hanning", n) * 40.0
    res = {}
    for name, split in (("one", None), ("two", "1")):
        monkeypatch.setenv("VBX_MFCC_CZT", "1")
        if split:
            monkeypatch.setenv("VBX_MFCC_CZT_SPLIT", split)
        v = pkg.VoxBox(0)
        monkeypatch.delenv("VBX_MFCC_CZT")
        if split:
            monkeypatch.delenv("VBX_MFCC_CZT_SPLIT")
        try:
            res[name] = v.mfcc(x, k, (lo, hi), sr)
        finally:
            v.close()
    for f in range(x.shape[0]):
        es, em = oracle.mfcc(x[f], k, lo, hi, sr)
        for name in res:
            assert res[name][1][f] == es
            assert np.all(rel_close(res[name][0][f], em, 1e-6)), (name, f, np.max(np.abs(res[name][0][f] - em)))
    assert np.all(rel_close(res["one"][0], res["two"][0], 1e-9))
    if n in (1103, 997, 2500):                                 # (4000 is split either way; 64 and 1601: top_bin > frame_len / 2, no split)
        assert not np.array_equal(res["one"][0], res["two"][0])


def test_mfcc_and_formants_odd_signals(vb, oracle):
    """Noise, tones, impulses, DC, silence, extreme scales through MFCC (matrix-core kernel at N = 1200) and
    find_formants: statuses exact (silence -> Err(LPC)), values within BASELINE tolerance of the oracle."""
    N, rng = 1200, np.random.default_rng(77)
    t = np.arange(N) / SR
    gens = [lambda: rng.standard_normal(N), lambda: np.sin(2 * np.pi * rng.uniform(60, 7000) * t),
            lambda: np.bincount(rng.integers(0, N, 5), minlength=N).astype(np.float64),
            lambda: 0.5 + 0.01 * rng.standard_normal(N), lambda: np.zeros(N),
            lambda: 1e-120 * rng.standard_normal(N), lambda: 1e100 * np.sin(2 * np.pi * 440 * t),
            lambda: np.sign(np.sin(2 * np.pi * 150 * t)) + 0.1 * rng.standard_normal(N)]
    X = np.array([gens[i % len(gens)]() for i in range(48)])
    w = oracle.window("hanning", N)
    m, ms = vb.mfcc(X * w, 13, (100.0, 8000.0), SR)
    for f in range(X.shape[0]):
        es, em = oracle.mfcc(X[f] * w, 13, 100.0, 8000.0, SR)
        assert ms[f] == es
        assert np.all(rel_close(m[f], em, 1e-6)), (f, np.max(np.abs(m[f] - em)))
    est0 = np.array([[fq, 1.0] for fq in (320.0, 1440.0, 2760.0, 3200.0)])
    out = vb.find_formants(X, SR, 12, est0, seg_start=np.arange(0, X.shape[0], 1))
    # EVERY signal class is compared, with a metric that follows the conditioning of the frame:
    #  * always: status exact, and every resonance the GPU reports is a root of the frame's
    #    LPC polynomial in the backward sense -- |P(z)| <= 1e-8 * sum |c_k| |z|^(p-k) at z = r e^(i theta) rebuilt from
    #    (frequency, bandwidth) (or at 1/conj(z) for a root that was reflected into the unit circle, src/spectrum.rs:171-174);
    #  * Burg coefficients within 1e-6 and formant Hz within 1e-4 wherever the ORACLE's own answer is stable: pure tones, impulses and 1e100 tones give a
    #    Burg polynomial with clustered roots that move by percents when the frame is perturbed by 1e-13 -- for those
    #    frames no implementation, the reference included, has digits to compare; the probe below detects them per frame.
    n_hz = {k: 0 for k in range(len(gens))}
    n_co = {k: 0 for k in range(len(gens))}
    n_cnt = {k: 0 for k in range(len(gens))}
    for f in range(X.shape[0]):
        es, ef, eres, eco = oracle.find_formants(X[f], SR, 12, est0)
        assert out["status"][f] == es, (f, out["status"][f], es)  # every frame its own segment: no carried state
        if es != 0:
            continue
        assert np.all(np.isfinite(out["formants"][f]))
        probe = X[f] * (1.0 + 1e-13 * rng.standard_normal(N))
        ps, pf, pres, pco = oracle.find_formants(probe, SR, 12, est0)
        # the resonance COUNT is compared on every frame -- the ill-conditioned classes too -- on which the oracle's own count
        # survives the 1e-13 perturbation (a cluster of near-multiple roots may still put one of them on either side of the
        # 50 Hz / Nyquist - 50 Hz filter of src/spectrum.rs:176-182: then not even the count has a reference value)
        e_cnt, p_cnt = int(np.sum(eres[:, 0] != 0.0)), int(np.sum(pres[:, 0] != 0.0))
        if ps == 0 and e_cnt == p_cnt:
            assert out["count"][f] == e_cnt, (f, out["count"][f], e_cnt)
            n_cnt[f % len(gens)] += 1
        if ps == 0 and np.all(rel_close(pco, eco, 1e-8)):        # the Burg recursion itself loses its digits on a pure tone
            assert np.all(rel_close(out["coeffs"][f], eco)), (f, np.max(np.abs(out["coeffs"][f] - eco)))
            n_co[f % len(gens)] += 1
        c = np.concatenate([[1.0], out["coeffs"][f]])            # P(x) = sum_k c_k x^(p-k)  (src/lib.rs:83-93)
        for fr, bw in out["res"][f, :int(out["count"][f])]:
            z = np.exp(-np.pi * bw / SR) * np.exp(2j * np.pi * fr / SR)
            be = min(abs(np.polyval(c, zz)) / np.polyval(np.abs(c), abs(zz)) for zz in (z, 1.0 / np.conj(z)))
            assert be <= 1e-8, (f, fr, bw, be)
        stable = ps == 0 and np.all(np.abs(pf[:, 0] - ef[:, 0]) <= 1e-7 * np.abs(ef[:, 0]) + 1e-12)
        if stable:
            assert np.all(np.abs(out["formants"][f, :, 0] - ef[:, 0]) <= 1e-4 * np.abs(ef[:, 0]) + 1e-9), f
            n_hz[f % len(gens)] += 1
    print("\nframes per signal class with Burg coefficients / formant Hz / resonance count compared:", n_co, n_hz, n_cnt)
    assert sum(n_cnt.values()) >= sum(n_hz.values())
    assert n_hz[0] == 6 and n_hz[3] == 6 and n_hz[7] == 6        # noise, DC + noise, square + noise: all well conditioned


def test_mfcc_bins_beyond_spectrum_is_panic_status(vb, oracle):
    x = np.ones((2, 64))
    m, st = vb.mfcc(x, 13, (100.0, 30000.0), 22050.0)      # mel points beyond the spectrum length
    es, _ = oracle.mfcc(x[0], 13, 100.0, 30000.0, 22050.0)
    assert es == oracle.ERR_PANIC and np.all(st == 4) and np.all(m == 0.0)


def test_zz_pitch_parity_report():
    """Runs last in this file: prints (and, on a GPU box, saves) the counters every _check_pitch call collected --
    how many top candidates took the tie escape, how many were voiced/unvoiced flips, how many strengths flipped."""
    import json
    import os
    tot = {k: sum(r[k] for r in PITCH_STATS) for k in ("frames", "n_cand", "n_flip", "n_top_bad", "n_top_swap", "n_vuv_flip",
                                                        "n_vuv_outside_tol")}
    tot["worst_tie_gap"] = max([r["worst_tie_gap"] for r in PITCH_STATS], default=0.0)
    print("\npitch parity counters:", json.dumps(tot))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "pitch_parity_stats.json"), "w") as fh:
            json.dump({"total": tot, "calls": PITCH_STATS}, fh, indent=1)
    assert tot["n_top_bad"] == 0 and tot["n_vuv_outside_tol"] == 0
    assert tot["n_top_swap"] <= max(1, tot["frames"] // 500), tot      # observed: 0
    assert tot["n_flip"] <= max(1, tot["n_cand"] // 1000), tot         # observed: 0


@pytest.mark.parametrize("n,hop,rect", [(1200, 480, False), (1200, 480, True), (1024, 512, False), (1024, 512, True),
                                        (2048, 1024, False), (2048, 1024, True), (1103, 441, False), (1102, 441, True),
                                        (512, 256, True), (513, 200, False), (704, 300, False), (901, 333, True), (1025, 400, False), (1199, 480, True),
                                        (1201, 480, False), (1600, 640, False), (2047, 900, True), (2049, 900, False), (3000, 1024, True),
                                        (4096, 2048, False), (4095, 2048, False)])
def test_pitch_fft_path_and_direct_path_agree(pkg, oracle, audio, monkeypatch, n, hop, rect):
    """Frame lengths 512..4096 take an FFT-based kernel (complex length 1024, 1200, 2048 or 4096, the frame zero padded);
    VBX_PITCH_MFMA=1 keeps the direct-sum (matrix-core) kernel that also serves every other frame length and the FFT path's
    fallback list.  The two produce the lag curve in different arithmetic (errors ~1e-16 S[0]): statuses and candidate
    COUNTS must be identical (the peak decisions are guarded by the fallback), values within the pitch tolerance.
    rect: rectangular frames, x[0] != 0 -- the fold seed (Q1) is then part of every lag."""
    rng = np.random.default_rng(5)
    F = pkg.frame_count(audio.size, n, hop)
    w = np.ones(n) if rect else oracle.window("hanning", n)
    x = _frames(audio, n, hop, list(range(0, F, max(3, F // 400)))) * w
    t = np.arange(n) / SR
    odd = np.array([rng.standard_normal(n), np.sin(2 * np.pi * 173.0 * t), np.sign(np.sin(2 * np.pi * 120 * t)),
                    np.bincount(rng.integers(0, n, 5), minlength=n).astype(np.float64), np.zeros(n),
                    0.5 + 0.01 * rng.standard_normal(n)]) * w
    X = np.concatenate([x, odd])
    res = {}
    for name, var in (("fft", None), ("direct", "VBX_PITCH_MFMA")):
        if var:
            monkeypatch.setenv(var, "1")
        v = pkg.VoxBox(0)
        if var:
            monkeypatch.delenv(var)
        try:
            res[name] = v.pitch(X, SR, 0.2, 75.0, 600.0, kmax=8)
        finally:
            v.close()
    (ca, ka, sa), (cb, kb, sb) = res["fft"], res["direct"]
    assert np.array_equal(ka, kb) and np.array_equal(sa, sb)
    ok = sa == 0
    nk = np.minimum(ka[ok], 8)
    A, B = ca[ok], cb[ok]
    assert np.all(np.abs(A[:, 0, 0] - B[:, 0, 0]) <= 1e-4 * np.abs(B[:, 0, 0]) + 1e-12)         # the PitchExtractor output
    dfreq = np.abs(np.sort(A[:, :, 0], axis=1) - np.sort(B[:, :, 0], axis=1))                      # the candidate sets
    assert np.all(dfreq <= 1e-4 * np.abs(np.sort(B[:, :, 0], axis=1)) + 1e-12)
    flips = np.abs(np.sort(A[:, :, 1], axis=1) - np.sort(B[:, :, 1], axis=1)) > 1e-4
    assert flips.sum() <= max(1, int(nk.sum()) // 100), int(flips.sum())
    assert not np.array_equal(A, B)                                                              # two kernels really ran


@pytest.mark.parametrize("n,hop", [(2048, 1024), (4096, 2048), (3000, 1200), (1600, 640), (4000, 1000)])
@pytest.mark.parametrize("kmax", [1, 8, 0])
def test_pitch_cut_lag_curve_changes_nothing(vb, pkg, oracle, audio, monkeypatch, n, hop, kmax):
    """The power-of-two FFT kernels keep only the lags the refinement can read (peak scan: [0, n/2]; sinc terms of candidates
    that pass the frequency filter: < 2 sample_rate / fmin + 4) -- pitch_curve_entries, vbx_pitch_refine.hpp.  With
    VBX_PITCH_CURVE_CUT=0 the whole curve is stored: every output must be BIT FOR BIT the same, for the settings that cut
    (fmin 75 Hz), for an fmin just high enough to cut at all, and for one so low that nothing is cut.  kmax 0: the whole Vec."""
    rng = np.random.default_rng(n + kmax)
    F = pkg.frame_count(audio.size, n, hop)
    w = oracle.window("hanning", n)
    x = _frames(audio, n, hop, list(range(0, F, max(3, F // 150)))) * w
    t = np.arange(n) / SR
    odd = np.array([rng.standard_normal(n), np.sin(2 * np.pi * 76.0 * t), np.sin(2 * np.pi * 30.0 * t), np.sign(np.sin(2 * np.pi * 80 * t)),
                    0.5 + 0.01 * rng.standard_normal(n)]) * w
    X = np.concatenate([x, odd])
    km = pkg.pitch_max_candidates(n) if kmax == 0 else kmax
    fmin_edge = 2.0 * SR / (n - 40)                   # the reach 2 ceil(sr / fmin) + 16 just below n
    res = {}
    for name, val in (("cut", None), ("whole", "0")):
        if val:
            monkeypatch.setenv("VBX_PITCH_CURVE_CUT", val)
        v = pkg.VoxBox(0)
        if val:
            monkeypatch.delenv("VBX_PITCH_CURVE_CUT")
        try:
            res[name] = [v.pitch(X, SR, 0.2, fmin, 600.0, kmax=km) for fmin in (75.0, fmin_edge, 20.0)]
        finally:
            v.close()
    for (ca, ka, sa), (cb, kb, sb) in zip(res["cut"], res["whole"]):
        assert np.array_equal(ka, kb) and np.array_equal(sa, sb) and np.array_equal(ca, cb)
    if kmax == 8:
        assert _check_pitch(vb, oracle, X[::4], SR, 0.2, 75.0, 600.0, 8, label=f"cut curve, N={n}") == 0


def test_pitch_fft_path_defers_undecidable_frames(vb, oracle):
    """Frames whose strict 3-point peak test lies inside the transforms' rounding error (a lag curve that is exactly zero
    or flat over a stretch: a few impulses, a constant) are not decided by the FFT kernel: they are listed and redone by
    the direct-sum kernel, so their candidate count is the oracle's.  Ordinary frames are not deferred."""
    N, rng = 1200, np.random.default_rng(11)
    w = oracle.window("hanning", N)
    ordinary = rng.standard_normal((6, N)) * w
    imp = np.zeros((4, N))
    for i in range(4):
        imp[i, rng.integers(100, 1100, 3)] = rng.uniform(0.5, 1.0, 3)
    cand, cnt, st = vb.pitch(ordinary, SR, 0.2, 75.0, 600.0, kmax=4)
    assert vb.last_unsure_count() == 0
    cand, cnt, st = vb.pitch(imp, SR, 0.2, 75.0, 600.0, kmax=4)
    assert vb.last_unsure_count() == 4
    for f in range(4):
        es, ec, en = oracle.pitch(imp[f], SR, 0.2, 75.0, 600.0)
        assert st[f] == es and cnt[f] == (en if es == 0 else 0), (f, st[f], es, cnt[f], en)
        if es == 0:
            k = min(4, en)
            assert np.all(np.abs(cand[f, :k, 0] - ec[:k, 0]) <= 1e-4 * np.abs(ec[:k, 0]) + 1e-12)


def test_pitch_fft_path_defers_frames_with_a_peak_on_the_frequency_bound(vb, oracle):
    """The frequency filter (src/periodic.rs:439) is a discrete decision on the curve as well: a peak whose parabolic
    frequency lies within the transforms' rounding error of fmin or fmax is not decided by the FFT kernel either.  A pure
    tone's lag-curve peaks give known parabolic frequencies; with fmax (or fmin) set EXACTLY on one of them the frame must be
    deferred, and with the bound a hair away it must not -- and in every case the count is the oracle's."""
    N = 1200
    w = oracle.window("hanning", N)
    t = np.arange(N) / SR
    x = (np.sin(2 * np.pi * 200.0 * t) * w)[None, :]
    # the parabolic frequencies of the frame's peaks, from the oracle's own lag curve
    y = _lag_curve(oracle, x[0])
    peaks = [k for k in range(1, N // 2 - 1) if y[k - 1] < y[k] > y[k + 1]]
    k = peaks[0]
    dr, d2r = 0.5 * (y[k + 1] - y[k - 1]), 2.0 * y[k] - (y[k - 1] - y[k + 1])
    f_par = SR / (k + dr / d2r)
    for fmin, fmax, deferred in ((75.0, f_par, 1), (f_par, 20000.0, 1), (75.0, f_par * (1 + 1e-9), 0), (75.0, f_par * (1 - 1e-9), 0), (75.0, 600.0, 0)):
        cand, cnt, st = vb.pitch(x, SR, 0.2, fmin, fmax, kmax=4)
        assert vb.last_unsure_count() == deferred, (fmin, fmax, vb.last_unsure_count())
        es, ec, en = oracle.pitch(x[0], SR, 0.2, fmin, fmax)
        assert st[0] == es and cnt[0] == en, (fmin, fmax, cnt[0], en)
