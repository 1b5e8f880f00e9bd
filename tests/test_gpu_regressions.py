"""Regression tests for defects found by review: each test fails on the code it was written against.

  * the time-sliced find_formants path left the tail of a longer last utterance untracked (vbx_api.hip run_find_formants)
  * laguerre() reduced the wave's highest degree with a shuffle butterfly from diverged code (k_roots.hip)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 48000.0


@pytest.mark.parametrize("delta", [+100, -100, 0])
def test_find_formants_time_sliced_uneven_last_utterance(vb, oracle, pkg, monkeypatch, delta):
    """>= 64 equal utterances of 64..383 frames and F >= 65536 take the time-sliced path (Burg / roots of slice j+1 beside
    the tracker of slice j).  A last utterance LONGER than the others (its end is n_frames, not a seg_start entry) has rows
    past seg_len that no slice tracks: those batches must take the unsliced path; a SHORTER last utterance stays sliced.
    Either way every row equals the unsliced scan bit for bit, and the last utterances equal the oracle's scan."""
    seg_len, n_seg = 256, 260
    N, H, P = 512, 160, 12
    F = seg_len * n_seg + delta if delta >= 0 else seg_len * n_seg + delta
    seg = np.arange(0, seg_len * n_seg, seg_len, dtype=np.int64)
    assert seg.size >= 64 and F >= 65536 and (n_seg - 1) * seg_len < F
    audio = vb.synth_speech((F - 1) * H + N, sample_offset=7 * 48000)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    monkeypatch.setenv("VBX_TRACKER_CHUNKED", "0")           # the time-sliced path (the default is the chunked scan since round 3)
    a = vb.find_formants(audio, SR, P, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F)
    monkeypatch.setenv("VBX_TRACKER_CHUNKED", "1")           # one launch each of Burg, roots, the chunked scan
    b = vb.find_formants(audio, SR, P, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F)
    monkeypatch.delenv("VBX_TRACKER_CHUNKED", raising=False)
    audio.free()
    assert np.array_equal(a["status"], b["status"])
    assert np.array_equal(a["res"], b["res"]) and np.array_equal(a["count"], b["count"])
    bad = np.nonzero(np.any(a["formants"] != b["formants"], axis=(1, 2)))[0]
    assert bad.size == 0, f"{bad.size} rows differ from the unsliced scan, first {bad[:5]} (F = {F}, last utterance {F - seg[-1]} frames)"
    # the oracle's sequential tracker over the GPU's resonance rows of the last two utterances (exact)
    for s in (n_seg - 2, n_seg - 1):
        est = est0.copy()
        end = F if s == n_seg - 1 else int(seg[s + 1])
        for t in range(int(seg[s]), end):
            if a["status"][t] == 0:
                est = oracle.estimate_formants(est, a["res"][t])
            assert np.array_equal(a["formants"][t], est), (s, t)


def _mixed_degree_polys(rng, n, length=13, deg_even=8, deg_odd=12):
    """One wave's worth (and a ragged second wave) of polynomials whose degrees interleave: lane 0 holds a LOW degree,
    lane 1 a constant (Err), even lanes degree 8, odd lanes degree 12 -- the wave's maximum is never in lane 0."""
    P = np.zeros((n, length), dtype=np.complex128)
    for f in range(n):
        lane = f % 64
        deg = 3 if lane == 0 else 0 if lane == 1 else deg_even if lane % 2 == 0 else deg_odd
        if lane == 5:
            deg = 2                                          # closed-form tail only: leaves the k loop at once
        if lane == 7:
            deg = 1
        P[f, :deg + 1] = rng.standard_normal(deg + 1)
        P[f, deg] = 1.0
    return P


def test_find_roots_mixed_degrees_in_one_wave(vb, oracle):
    """laguerre() runs under divergence (lanes leave find_roots_emit early or loop a different number of times): the
    wave's highest degree must come from the active lanes only.  With a shuffle butterfly, lane 0 (degree 3) never saw
    the odd lanes' degree 12 once lane 1 (a constant: Err) had left, and their Horner chains started too low."""
    rng = np.random.default_rng(31)
    P = _mixed_degree_polys(rng, 64 + 37)
    got, st = vb.find_roots(P)
    for f in range(P.shape[0]):
        es, er = oracle.find_roots_mut(P[f])
        assert st[f] == es, (f, st[f], es)
        if es != 0:
            continue
        g = got[f].copy()
        tol = 1e-7 * np.maximum(1.0, np.abs(er))
        if not np.all(np.abs(g - er) <= tol):                # only the conjugate pair of the quadratic tail may swap
            nz = int(np.max(np.nonzero(er)[0])) if np.any(er != 0) else 0
            g[[nz - 1, nz]] = g[[nz, nz - 1]]
        assert np.all(np.abs(g - er) <= tol), (f, g, er)


def test_find_roots_f32_mixed_degrees_in_one_wave(vb, oracle):
    rng = np.random.default_rng(32)
    P = _mixed_degree_polys(rng, 64, length=10, deg_even=6, deg_odd=9).astype(np.complex64)
    P[P.real != 0] = np.clip(P[P.real != 0].real, -1.0, 1.0)
    for f in range(64):                                      # keep the leading coefficients at 1 after clipping
        nz = np.nonzero(P[f])[0]
        P[f, nz[-1]] = 1.0
    got, st = vb.find_roots_f32(P)
    for f in range(P.shape[0]):
        es, er = oracle.find_roots_f32(P[f])
        assert st[f] == es, (f, st[f], es)
        if es != 0:
            continue
        k = er.size
        pv = np.polyval(P[f, ::-1].astype(np.complex128), got[f, :k].astype(np.complex128))
        scale = np.polyval(np.abs(P[f, ::-1]).astype(np.float64), np.abs(got[f, :k]).astype(np.float64))
        ok = np.abs(pv) <= 5e-4 * scale
        assert np.all(ok | (got[f, :k] == 0)), (f, np.abs(pv) / scale)
