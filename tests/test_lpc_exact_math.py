"""The round-6 LPC path as MATH, on the CPU (tests/lpc_exact_model.py = numpy model of k_lpc.hip's conditioning probe and
k_lpc_exact.hip's double-double redo), on the material that needs it: frames of the real 44.1 kHz fixture.

What is pinned here, without a GPU:
  * the premise: on oversampled speech the REFERENCE's f64 row (sequential fold src/periodic.rs:279-288 + recursion
    src/spectrum.rs:63-84, restated by the oracle) is further than 1e-6 (parity metric) from the same recursion in long double on
    long-double lag sums, on some rows -- no f64 evaluation can be held to it there;
  * the double-double path is the exact row: within 1e-9 of the long-double arbiter on every row it is run on;
  * the probe is sufficient: a row it does NOT list is within 1e-6 of the arbiter when computed in plain f64 from numpy's (pairwise)
    lag sums -- so the returned rows are within 1e-6 of the exact row everywhere, which is what tests/test_gpu_lpc_exact.py and the
    soak then hold the GPU to."""
import os
import wave

import numpy as np

from lpc_exact_model import lag_sums_dd, levinson, levinson_dd, lpc_rows, parity_metric


def _speech_frames(oracle, golden_dir, n, hop, F, gain=0.83):
    with wave.open(os.path.join(golden_dir, "sample-two_vowels.wav"), "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float64) / 32767.0
    rng = np.random.default_rng(44)
    x = gain * pcm + 10 ** (-70 / 20) * (2 * rng.random(pcm.size) - 1)          # the soak's recipe: a gain and a -70 dB dither
    win = oracle.window("hanning", n)
    idx = np.arange(F)[:, None] * hop + np.arange(n)[None, :]
    return x[idx] * win[None, :]


def _arbiter(xw, p):
    xl = xw.astype(np.longdouble)
    n = xl.shape[1]
    r = np.stack([xl[:, 0] + np.sum(xl[:, 1:n - k] * xl[:, 1 + k:n], axis=1) for k in range(p + 1)], axis=1)
    return levinson(r, np.longdouble)


def test_the_reference_rows_are_not_the_exact_rows_on_speech_and_the_model_rows_are(oracle, golden_dir):
    n, hop, F = 1103, 441, 250
    for p in (12, 13):
        xw = _speech_frames(oracle, golden_dir, n, hop, F)
        al = _arbiter(xw, p)
        ref = np.stack([oracle.lpc(oracle.autocorrelate(f, p + 1), p) for f in xw])
        d_ref = parity_metric(ref, al)
        assert d_ref.max() > 2e-6, d_ref.max()                                  # the premise
        got, listed = lpc_rows(xw, p)
        d = parity_metric(got, al)
        assert 0 < listed.sum() < F // 2, listed.sum()                          # some rows, not most
        assert d.max() <= 1e-6, (d.max(), d_ref.max())                          # every returned row: within 1e-6 of the exact row
        assert d[listed].max() <= 1e-9, d[listed].max()                         # the redone rows ARE the exact rows
        # every row the reference misses by more than 2e-6 is one the probe lists (its own error is what the probe measures)
        assert np.all(listed[d_ref > 2e-6] | (d[d_ref > 2e-6] <= 1e-6))


def test_double_double_rows_on_well_conditioned_frames_are_the_f64_rows(oracle):
    """white noise: nothing is ill-conditioned, the double-double row rounded once and the f64 row agree to a few ulp of the row"""
    rng = np.random.default_rng(5)
    n, p, F = 512, 12, 40
    xw = rng.standard_normal((F, n)) * oracle.window("hanning", n)[None, :]
    rh, rl = lag_sums_dd(xw, p)
    a_dd = levinson_dd(rh, rl)
    al = _arbiter(xw, p)
    assert parity_metric(a_dd, al).max() <= 1e-13
    got, listed = lpc_rows(xw, p)
    assert not listed.any() and parity_metric(got, al).max() <= 1e-10
