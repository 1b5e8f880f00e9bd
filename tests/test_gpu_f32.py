"""Sample = f32 instantiation of the slice traits: vbx_autocorrelate_f32, vbx_normalize_f32,
vbx_lpc_mut_f32, vbx_autocorr_lpc_f32, vbx_lpc_burg_f32, vbx_mfcc_f32.

The reference computes these in f32 (generic code monomorphised at T = f32); none of its tests does, so the f32
restatement in oracle/vbx_oracle_f32.c is parity-unpinned.  The library widens on load, computes in f64 and rounds each
result to f32 once.  Two checks per entry point:
  * identity: the f32 entry point == the f64 entry point on the widened frames, rounded to f32 -- bit for bit (they are
    the same kernels instantiated at another Sample type);
  * accuracy: with e_ref = |oracle_f32 - oracle_f64| (the reference's own f32 rounding) and e_gpu = |gpu_f32 - oracle_f64|,
    e_gpu <= e_ref + one f32 ulp of the row's scale: the library is at least as close to the exact answer as the
    reference's f32 arithmetic is, hence within 2 e_ref + ulp of the f32 restatement.
Needs a real MI355X."""
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


@pytest.mark.parametrize("n,lags", [(512, 13), (1200, 13), (256, 9), (700, 40), (1200, 1200)])   # few-lag and matrix-core paths
def test_autocorrelate_f32(vb, oracle, audio, n, lags):
    x = _frames32(audio, n, 211, 12, oracle.window("hanning", n))
    x[3] = np.random.default_rng(n).uniform(-1, 1, n).astype(np.float32)         # rectangular frame: x[0] != 0 (Q1)
    got = vb.autocorrelate_f32(x, lags)
    assert got.dtype == np.float32
    same = vb.autocorrelate(x.astype(np.float64), lags).astype(np.float32)
    assert np.array_equal(got, same)
    r32 = np.stack([oracle.autocorrelate_f32(f, lags) for f in x])
    r64 = np.stack([oracle.autocorrelate(f.astype(np.float64), lags) for f in x])
    e_gpu, e_ref = _accuracy(got, r32, r64, "autocorrelate")
    assert e_gpu <= 1.0 * ULP and e_ref > e_gpu                                   # one rounding vs n of them


def test_autocorrelate_f32_applies_the_window_in_f32(vb, oracle, audio):
    n = 512
    w = oracle.window("hanning", n).astype(np.float32)
    raw = np.stack([audio[t * 300:t * 300 + n] for t in range(6)]).astype(np.float32)
    got = vb.autocorrelate_f32(raw, 13, window=w)
    assert np.array_equal(got, vb.autocorrelate_f32((raw * w).astype(np.float32), 13))


def test_normalize_and_lpc_f32(vb, oracle, audio):
    n, p = 512, 12
    x = _frames32(audio, n, 333, 10, oracle.window("hanning", n))
    r = vb.autocorrelate_f32(x, p + 1)
    rn = vb.normalize_f32(r)
    for f in range(r.shape[0]):
        assert np.all(np.abs(rn[f] - oracle.normalize_f32(r[f])) <= ULP * np.abs(rn[f]) + 1e-38), f
    ac, kc = vb.lpc_mut_f32(rn, p)
    assert ac.dtype == np.float32 and np.all(ac[:, 0] == 1.0)
    ac64, kc64 = vb.lpc_mut(rn.astype(np.float64), p)
    assert np.array_equal(ac, ac64.astype(np.float32)) and np.array_equal(kc, kc64.astype(np.float32))
    a32 = np.stack([oracle.lpc_f32(row, p)[0] for row in rn])
    a64 = np.stack([oracle.lpc(row.astype(np.float64), p) for row in rn])
    print("\nlpc_f32 max rel error: gpu %.2e, f32 restatement %.2e" % _accuracy(ac, a32, a64, "lpc"))
    # fused autocorrelate -> normalize -> lpc
    r2, a2 = vb.autocorr_lpc_f32(x, p, normalize=True)
    r2d, a2d = vb.autocorr_lpc(x.astype(np.float64), p, normalize=True)
    assert np.array_equal(r2, r2d.astype(np.float32)) and np.array_equal(a2, a2d.astype(np.float32))


@pytest.mark.parametrize("n,p", [(512, 12), (1200, 12), (256, 8), (2000, 16)])
def test_lpc_praat_f32(vb, oracle, audio, n, p):
    x = _frames32(audio, n, 401, 9, oracle.window("hanning", n))
    x[8] = 0.0                                                                    # all-zero frame -> Err(LPC)
    co, st = vb.lpc_praat_f32(x, p)
    assert co.dtype == np.float32
    co64, st64 = vb.lpc_praat(x.astype(np.float64), p)
    assert np.array_equal(st, st64) and np.array_equal(co, co64.astype(np.float32))
    for f in range(x.shape[0]):
        es, ec = oracle.lpc_burg_f32(x[f], p)
        assert st[f] == es, f
    ok = st == 0
    c32 = np.stack([oracle.lpc_burg_f32(f, p)[1] for f in x[ok]])
    c64 = np.stack([oracle.lpc_burg(f.astype(np.float64), p)[1] for f in x[ok]])
    print("\nlpc_praat_f32 n=%d max rel error: gpu %.2e, f32 restatement %.2e" % ((n,) + _accuracy(co[ok], c32, c64, "burg")))
    assert np.all(co[~ok] == 0.0)


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


@pytest.mark.parametrize("n,hop", [(1200, 480), (1024, 512), (2048, 1024), (1103, 441)])     # one per FFT plan, and a padded odd length
def test_pitch_f32(vb, oracle, audio, n, hop):
    """Pitched<f32, f32>::pitch: identity with the f64 path on the widened frames (rounded once), and against the f32
    restatement -- whose lag curve carries the rounding of 1200 f32 folds, so candidate COUNTS may differ on peaks of
    that size (counted, bounded: more of them the longer the frame); the top candidate of voiced frames agrees within 1e-4 / 1e-3."""
    F = min(60, (audio.size - n) // hop + 1)
    x = _frames32(audio, n, hop, F, oracle.window("hanning", n))
    cand, cnt, st = vb.pitch_f32(x, SR, 0.2, 75.0, 600.0, kmax=4)
    assert cand.dtype == np.float32 and np.all(st == 0)
    c64, k64, s64 = vb.pitch(x.astype(np.float64), SR, 0.2, 75.0, 600.0, kmax=4)
    assert np.array_equal(cand, c64.astype(np.float32)) and np.array_equal(cnt, k64) and np.array_equal(st, s64)
    n_count_diff = n_voiced = 0
    for f in range(F):
        es, ec, en = oracle.pitch_f32(x[f], SR, 0.2, 75.0, 600.0)
        assert es == 0
        n_count_diff += int(en != cnt[f])
        if ec[0, 0] > 0 and (en == 1 or ec[0, 1] - ec[1, 1] > 1e-2):      # clearly voiced: the decision is not a near tie
            n_voiced += 1
            assert abs(cand[f, 0, 0] - ec[0, 0]) <= 1e-4 * ec[0, 0] and abs(cand[f, 0, 1] - ec[0, 1]) <= 1e-3, (f, cand[f, 0], ec[0])
    print("\npitch_f32: %d clearly voiced frames compared, candidate count differs from the f32 restatement in %d of %d frames"
          % (n_voiced, n_count_diff, F))
    assert n_voiced >= F // 3 and n_count_diff <= F // 4
