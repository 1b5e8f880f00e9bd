"""Sample = f32 instantiation of the slice traits: vbx_autocorrelate_f32, vbx_normalize_f32,
vbx_lpc_mut_f32, vbx_autocorr_lpc_f32, vbx_lpc_burg_f32, vbx_pitch_f32 (reference-faithful) and their *_f32_wide forms.

The reference computes these in f32 (generic code monomorphised at T = f32); none of its tests does, so the f32
restatement in oracle/vbx_oracle_f32.c is parity-unpinned.  Two forms, two kinds of check:
  * the plain names run every fold in f32 in the reference's order (k_f32.hip): EQUAL, bit for bit, to the f32 restatement
    (lag sums, normalised rows, Levinson and Burg coefficients, statuses, pitch candidate COUNTS; the pitch candidates
    themselves within the Brent iteration's scatter, their f32 values usually equal);
  * the *_wide names widen on load, compute in f64 and round once: identical to the f64 entry points on the widened frames
    (rounded to f32), and at least as close to the exact (f64) answer as the reference's f32 arithmetic is.
MFCC exists only in the wide form (rustfft's f32 arithmetic is not in the tree).  Needs a real MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SR = 48000.0
ULP = float(np.finfo(np.float32).eps)


def _frames32(audio, n, hop, count, window):
    x = np.stack([audio[t * hop:t * hop + n] for t in range(count)]).astype(np.float32)
    return (x * window.astype(np.float32)).astype(np.float32)            # Windower<f32>: the product is an f32


@pytest.fixture(scope="module")
def audio(pkg):
    import __graft_entry__ as g
    synth = __import__("importlib").import_module(g.PKG_NAME + ".synth")
    return synth.synth_speech(48000 + 4096, sample_offset=48000)


def _accuracy(gpu, ref32, ref64, what):
    scale = np.max(np.abs(ref64), axis=-1, keepdims=True)
    e_gpu = np.abs(gpu.astype(np.float64) - ref64)
    e_ref = np.abs(ref32.astype(np.float64) - ref64)
    worst_ref = np.max(e_ref, axis=-1, keepdims=True)
    assert np.all(e_gpu <= worst_ref + 1.5 * ULP * scale), (what, float(np.max(e_gpu / scale)), float(np.max(e_ref / scale)))
    return float(np.max(e_gpu / scale)), float(np.max(e_ref / scale))


@pytest.mark.parametrize("n,lags", [(512, 13), (1200, 13), (256, 9), (700, 40), (1200, 1200), (65, 65), (4096, 100)])
def test_autocorrelate_f32_equals_the_f32_fold(vb, oracle, audio, n, lags):
    """impl<T: Sample> Autocorrelate<T> for [T] at T = f32 (src/periodic.rs:276-289): every lag a sequential f32 fold seeded
    with x[0] -- bit for bit."""
    x = _frames32(audio, n, 211 if n <= 1200 else 1, 12 if n <= 1200 else 3, oracle.window("hanning", n))
    x[1] = np.random.default_rng(n).uniform(-1, 1, n).astype(np.float32)         # rectangular frame: x[0] != 0 (Q1)
    got = vb.autocorrelate_f32(x, lags)
    assert got.dtype == np.float32
    r32 = np.stack([oracle.autocorrelate_f32(f, lags) for f in x])
    assert np.array_equal(got, r32)
    # the wide form: the f64 kernels on the widened frame, one rounding -- closer to the exact sums than the f32 folds are
    wide = vb.autocorrelate_f32(x, lags, wide=True)
    assert np.array_equal(wide, vb.autocorrelate(x.astype(np.float64), lags).astype(np.float32))
    r64 = np.stack([oracle.autocorrelate(f.astype(np.float64), lags) for f in x])
    e_gpu, e_ref = _accuracy(wide, r32, r64, "autocorrelate")
    assert e_gpu <= 1.0 * ULP and e_ref > e_gpu                                   # one rounding vs n of them


def test_autocorrelate_f32_applies_the_window_in_f32(vb, oracle, audio):
    n = 512
    w = oracle.window("hanning", n).astype(np.float32)
    raw = np.stack([audio[t * 300:t * 300 + n] for t in range(6)]).astype(np.float32)
    for wide in (False, True):
        got = vb.autocorrelate_f32(raw, 13, window=w, wide=wide)
        assert np.array_equal(got, vb.autocorrelate_f32((raw * w).astype(np.float32), 13, wide=wide))


def test_normalize_and_lpc_f32(vb, oracle, audio):
    n, p = 512, 12
    x = _frames32(audio, n, 333, 10, oracle.window("hanning", n))
    r = vb.autocorrelate_f32(x, p + 1)
    rn = vb.normalize_f32(r)
    assert np.array_equal(rn, np.stack([oracle.normalize_f32(row) for row in r]))
    ac, kc = vb.lpc_mut_f32(rn, p)
    assert ac.dtype == np.float32 and np.all(ac[:, 0] == 1.0)
    for f in range(rn.shape[0]):
        ea, ek = oracle.lpc_f32(rn[f], p)
        assert np.array_equal(ac[f], ea) and np.array_equal(kc[f], ek), f
    # fused autocorrelate -> normalize -> lpc, each step the f32 statement
    r2, a2 = vb.autocorr_lpc_f32(x, p, normalize=True)
    assert np.array_equal(r2, rn) and np.array_equal(a2, ac)
    r3, a3 = vb.autocorr_lpc_f32(x, p, normalize=False)
    assert np.array_equal(r3, r) and np.array_equal(a3, np.stack([oracle.lpc_f32(row, p)[0] for row in r]))
    # wide forms: the f64 kernels, rounded once
    acw, kcw = vb.lpc_mut_f32(rn, p, wide=True)
    ac64, kc64 = vb.lpc_mut(rn.astype(np.float64), p)
    assert np.array_equal(acw, ac64.astype(np.float32)) and np.array_equal(kcw, kc64.astype(np.float32))
    a64 = np.stack([oracle.lpc(row.astype(np.float64), p) for row in rn])
    print("\nlpc_f32_wide max rel error: gpu %.2e, f32 restatement %.2e" % _accuracy(acw, ac, a64, "lpc"))
    r2w, a2w = vb.autocorr_lpc_f32(x, p, normalize=True, wide=True)
    r2d, a2d = vb.autocorr_lpc(x.astype(np.float64), p, normalize=True)
    assert np.array_equal(r2w, r2d.astype(np.float32)) and np.array_equal(a2w, a2d.astype(np.float32))


@pytest.mark.parametrize("n,p", [(512, 12), (1200, 12), (256, 8), (2000, 16), (70, 5)])
def test_lpc_praat_f32_equals_the_f32_recursion(vb, oracle, audio, n, p):
    """LPC::lpc_praat_mut at T = f32 (src/spectrum.rs:101-146): the two sums of every order as sequential f32 folds."""
    F = 70 if n <= 512 else 9                                                     # more than one wavefront of lanes at the short lengths
    x = _frames32(audio, n, 401 if n > 512 else 97, F, oracle.window("hanning", n))
    x[F - 1] = 0.0                                                                # all-zero frame -> Err(LPC)
    co, st = vb.lpc_praat_f32(x, p)
    assert co.dtype == np.float32
    for f in range(F):
        es, ec = oracle.lpc_burg_f32(x[f], p)
        assert st[f] == es, f
        if es == 0:
            assert np.array_equal(co[f], ec), (f, co[f], ec)
    assert st[F - 1] == 1 and np.all(co[F - 1] == 0.0)
    cow, stw = vb.lpc_praat_f32(x, p, wide=True)
    co64, st64 = vb.lpc_praat(x.astype(np.float64), p)
    assert np.array_equal(stw, st64) and np.array_equal(cow, co64.astype(np.float32))
    ok = st == 0
    c64 = np.stack([oracle.lpc_burg(f.astype(np.float64), p)[1] for f in x[ok]])
    print("\nlpc_praat_f32_wide n=%d max rel error: gpu %.2e, f32 recursion %.2e" % ((n,) + _accuracy(cow[ok], co[ok], c64, "burg")))


@pytest.mark.parametrize("n", [1200, 512, 400])        # matrix-core two-stage DFT, two-stage DFT, and shapes it declines
def test_mfcc_f32(vb, oracle, audio, n):
    x = _frames32(audio, n, 389, 8, oracle.window("hanning", n))
    m, st = vb.mfcc_f32(x, 13, (100.0, 8000.0), SR)
    assert m.dtype == np.float32 and np.all(st == 0)
    m64, st64 = vb.mfcc(x.astype(np.float64), 13, (100.0, 8000.0), SR)
    assert np.array_equal(m, m64.astype(np.float32))
    m32 = np.stack([oracle.mfcc_f32(f, 13, 100.0, 8000.0, SR)[1] for f in x])
    r64 = np.stack([oracle.mfcc(f.astype(np.float64), 13, 100.0, 8000.0, SR)[1] for f in x])
    print("\nmfcc_f32 n=%d max rel error: gpu %.2e, f32 restatement %.2e" % ((n,) + _accuracy(m, m32, r64, "mfcc")))
    bad, stb = vb.mfcc_f32(np.ones((2, 64), np.float32), 13, (100.0, 30000.0), 22050.0)     # bins beyond the spectrum
    assert np.all(stb == 4) and np.all(bad == 0.0)


@pytest.mark.parametrize("n,hop", [(1200, 480), (1024, 512), (2048, 1024), (1103, 441), (300, 100)])
def test_pitch_f32_follows_the_f32_lag_curve(vb, oracle, audio, n, hop):
    """Pitched<f32, f32>::pitch: the lag curve is built in f32 exactly as the reference builds it (autocorrelate ->
    normalize -> / lag window), so the candidate COUNT of every frame equals the f32 restatement's; the candidates go
    through the shared Brent / sinc refinement (f64 on the widened curve, T = f32 roundings): frequencies within 1e-4,
    strengths within 1e-4, and as f32 values usually the same bits."""
    F = min(60, (audio.size - n) // hop + 1)
    x = _frames32(audio, n, hop, F, oracle.window("hanning", n))
    cand, cnt, st = vb.pitch_f32(x, SR, 0.2, 75.0, 600.0, kmax=4)
    assert cand.dtype == np.float32
    n_same_bits = n_cmp = 0
    for f in range(F):
        es, ec, en = oracle.pitch_f32(x[f], SR, 0.2, 75.0, 600.0)
        assert st[f] == es and cnt[f] == (en if es == 0 else 0), (f, st[f], es, cnt[f], en)
        if es != 0:
            continue
        tie = en > 1 and abs(ec[0, 1] - ec[1, 1]) < 1e-3
        if not tie:
            assert abs(cand[f, 0, 0] - ec[0, 0]) <= 1e-4 * abs(ec[0, 0]) + 1e-12 and abs(cand[f, 0, 1] - ec[0, 1]) <= 1e-4, (f, cand[f, 0], ec[0])
            n_cmp += 1
            n_same_bits += int(np.float32(ec[0, 0]) == cand[f, 0, 0] and np.float32(ec[0, 1]) == cand[f, 0, 1])
    print("\npitch_f32 n=%d: %d top candidates compared, %d with identical f32 bits" % (n, n_cmp, n_same_bits))
    assert n_cmp >= F // 2
    # the wide form: the f64 path on the widened frames, rounded once
    cw, kw, sw = vb.pitch_f32(x, SR, 0.2, 75.0, 600.0, kmax=4, wide=True)
    c64, k64, s64 = vb.pitch(x.astype(np.float64), SR, 0.2, 75.0, 600.0, kmax=4)
    assert np.array_equal(cw, c64.astype(np.float32)) and np.array_equal(kw, k64) and np.array_equal(sw, s64)


def test_pitch_f32_whole_list_and_odd_frames(vb, oracle, audio):
    """The whole f32 candidate Vec (counts and every candidate against the f32 restatement), silence and a rectangular frame."""
    n = 1200
    x = _frames32(audio, n, 480, 12, oracle.window("hanning", n))
    x[3] = 0.0                                                                    # silence: 0 / 0 lag curve -> NaN strengths
    x[5] = np.random.default_rng(5).uniform(-1, 1, n).astype(np.float32)          # rectangular noise: many candidates
    kfull = n // 4 + 2
    cand, cnt, st = vb.pitch_f32(x, SR, 0.2, 75.0, 600.0, kmax=kfull)
    for f in range(x.shape[0]):
        es, ec, en = oracle.pitch_f32(x[f], SR, 0.2, 75.0, 600.0)
        assert st[f] == es and cnt[f] == (en if es == 0 else 0), (f, st[f], es, cnt[f], en)
        if es == 0:
            g = cand[f, :en].astype(np.float64)
            g = g[np.argsort(g[:, 0], kind="stable")]
            e = ec[:en][np.argsort(ec[:en, 0], kind="stable")]
            assert np.all(np.abs(g[:, 0] - e[:, 0]) <= 1e-4 * np.abs(e[:, 0]) + 1e-12), f
            assert np.mean(np.abs(g[:, 1] - e[:, 1]) <= 1e-4) >= 0.98, f
            assert np.all(np.diff(cand[f, :en, 1]) <= 0.0), f
