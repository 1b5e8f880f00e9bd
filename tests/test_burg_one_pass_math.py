"""The one-pass Burg recursion (k_burg_fast.hip) as MATH, on the CPU: its numpy model (tests/burg_one_pass_model.py)
against the oracle's direct recursion (oracle/vbx_oracle.c vbxo_lpc_burg = src/spectrum.rs:101-146).

What is pinned here:
  * the recursion is the reference's: on speech-like frames it agrees with the direct sums to ~1e-11 of the row's largest
    coefficient (the lag sums' own rounding, amplified by the frame's conditioning);
  * the guard the kernel evaluates is sufficient: every frame it trusts is inside 1e-6 in the parity metric, and frames
    where the recursion is wrong by more than that (pure tones, DC, silence, NaN) are never trusted.
"""
import importlib

import numpy as np

from burg_one_pass_model import adversarial_frames, burg_one_pass, parity_metric

P = 12


def _oracle_rows(oracle, X):
    st = np.zeros(X.shape[0], dtype=np.int32)
    co = np.zeros((X.shape[0], P))
    for f in range(X.shape[0]):
        st[f], co[f] = oracle.lpc_burg(X[f], P)
    return st, co


def _speech_frames(pkg, oracle, N, H, F):
    synth = importlib.import_module(pkg.__name__ + ".synth")
    audio = synth.synth_speech((F - 1) * H + N, 5 * 48000)
    w = oracle.window("hanning_periodic", N)
    idx = np.arange(F)[:, None] * H + np.arange(N)[None, :]
    return audio[idx] * w


def test_recursion_equals_the_direct_sums_on_speech(pkg, oracle):
    for N, H in ((512, 512), (1200, 480), (257, 100)):
        X = _speech_frames(pkg, oracle, N, H, 1500)
        st, exp = _oracle_rows(oracle, X)
        co, trusted = burg_one_pass(X, P)
        assert np.all(st == 0)
        scale = np.max(np.abs(exp), axis=1)
        absolute = np.max(np.abs(co - exp), axis=1) / scale
        assert absolute.max() < 1e-9, (N, absolute.max())                 # observed ~1e-11
        m = parity_metric(co, exp)
        assert m[trusted].max() < 1e-7, (N, m[trusted].max())             # the guard's target is 5e-7 as a BOUND
        assert trusted.mean() > 0.9, (N, trusted.mean())                  # ~1-2 % go to the direct recursion


def test_recursion_at_the_other_orders(pkg, oracle):
    X = _speech_frames(pkg, oracle, 512, 512, 600)
    for order in (8, 10, 13, 16):
        exp = np.array([oracle.lpc_burg(x, order)[1] for x in X])
        co, trusted = burg_one_pass(X, order)
        m = parity_metric(co, exp)
        assert m[trusted].max() < 1e-7 and trusted.mean() > 0.85, (order, m[trusted].max(), trusted.mean())


def test_guard_turns_away_what_the_recursion_cannot_do(oracle):
    rng = np.random.default_rng(11)
    for N in (512, 1200):
        w = oracle.window("hanning_periodic", N)
        X = adversarial_frames(N, rng, count=240) * w
        st, exp = _oracle_rows(oracle, X)
        co, trusted = burg_one_pass(X, P)
        assert not np.any(trusted & (st != 0))                            # Err(LPC) and NaN frames are never trusted
        m = parity_metric(co[trusted], exp[trusted])
        assert m.max() < 1e-6, (N, m.max())
        wrong = (st == 0) & ~(parity_metric(co, exp) < 1e-6)              # the recursion alone would fail here
        assert wrong.sum() > 20 and not np.any(wrong & trusted), (N, wrong.sum())
