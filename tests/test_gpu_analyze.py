"""vbx_analyze_frames_f64 (the user's whole frame loop as one call, writing per-frame records) and the RCCL record
gather of the library, on a real MI355X.  The records are checked against the separate entry points (same
tolerances as their own parity tests) and against the CPU oracle."""
import numpy as np
import pytest

from conftest import rel_close

pytestmark = pytest.mark.gpu

SR, N, H, P = 48000.0, 1200, 480, 12


@pytest.fixture(scope="module")
def audio_d(vb):
    d = vb.synth_speech(6 * 48000, sample_offset=2 * 48000)     # voiced glide + one unvoiced second
    yield d
    d.free()


def _oracle_records(oracle, pkg, audio, F, seg, N=N, H=H):
    w = oracle.window("hanning", N)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    rec = np.zeros((F, 36))
    st = np.zeros((3, F), dtype=np.int32)
    est = est0.copy()
    for t in range(F):
        fr = audio[t * H:t * H + N]
        xw = fr * w
        s, c, _ = oracle.pitch(xw, SR, 0.2, 75.0, 600.0, cap=1)
        st[0, t] = s
        rec[t, 0:2] = c[0]
        if t in seg:
            est = est0.copy()
        s, est, _, _ = oracle.find_formants(fr, SR, P, est)
        st[1, t] = s
        rec[t, 2:10] = est.reshape(-1)
        s, m = oracle.mfcc(xw, 13, 100.0, 8000.0, SR)
        st[2, t] = s
        rec[t, 10:23] = m
        rec[t, 23:36] = oracle.lpc(oracle.autocorrelate(xw, P + 1), P)
    return rec, st


# (1200, 480), (1024, 512), (2048, 1024): pitch + LPC + MFCC from one FFT of the frame; (512, 256): the same from the zero-padded
# frame (512 divides the 1024 plan's 2048 points: its DFT bins are every 4th bin); (800, 320), (600, 240): likewise in the
# 1200 plan (2400 = 3 * 800 = 4 * 600) instead of the 1024 plan their length would pick; (1103, 441), (1199, 480), (1600, 640),
# (960, 480), (700, 350), (3000, 1200), (4000, 2000): lengths that do not divide their transform -- MFCC's bins interpolated from
# the transform's inside the fused kernel (round 5; one instance per plan), held here to the chirp-z kernel of vbx_mfcc_f64 and to
# the oracle; (4096, 2048): complex FFT of 4096; (256, 128): no fused kernel
@pytest.mark.parametrize("N,H", [(1200, 480), (1024, 512), (2048, 1024), (1103, 441), (1600, 640), (512, 256), (4096, 2048), (256, 128),
                                 (800, 320), (600, 240), (960, 480), (1199, 480), (700, 350), (3000, 1200), (4000, 2000)])
def test_analyze_frames_matches_the_oracle_and_the_separate_entry_points(vb, pkg, oracle, audio_d, N, H):
    audio = audio_d.numpy()
    F = pkg.frame_count(audio.size, N, H)
    q = min(F // 4, 150)
    seg = np.array([0, q, q + 1, min(400, F - 10)], dtype=np.int64)              # a one-frame utterance among them
    params = pkg.AnalysisParams.make(SR)
    assert params.columns() == {"pitch": (0, 2), "formants": (2, 8), "mfcc": (10, 13), "lpc": (23, 13)}
    rec, st = vb.analyze_frames(audio_d, params, seg_start=seg, frame_len=N, stride=H, n_frames=F)
    assert rec.shape == (F, 36) and st.shape == (3, F)
    # the separate entry points on the same frames
    han = vb.window(pkg.WINDOW_HANNING, N)
    cand, cnt, pst = vb.pitch(audio_d, SR, 0.2, 75.0, 600.0, kmax=1, frame_len=N, stride=H, n_frames=F, window=han)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    ff = vb.find_formants(audio_d, SR, P, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F)
    mf, mst = vb.mfcc(audio_d, 13, (100.0, 8000.0), SR, frame_len=N, stride=H, n_frames=F, window=han)
    r, a = vb.autocorr_lpc(audio_d, P, frame_len=N, stride=H, n_frames=F, window=han)
    assert np.array_equal(st[0], pst) and np.array_equal(st[1], ff["status"]) and np.array_equal(st[2], mst)
    # pitch: same refinement code -> same top candidate up to the documented Brent sensitivity
    assert np.all(np.abs(rec[:, 0] - cand[:, 0, 0]) <= 1e-4 * np.abs(cand[:, 0, 0]))
    assert np.array_equal(rec[:, 2:10], ff["formants"].reshape(F, 8))        # same kernels, same stream order: bit-equal
    assert np.all(rel_close(rec[:, 10:23], mf)) and np.all(rel_close(rec[:, 23:36], a))
    # and the oracle on a subset (it takes ~10 ms per frame)
    sub = 160 if N <= 1200 else min(154, F)
    orec, ost = _oracle_records(oracle, pkg, audio, sub, set(seg.tolist()), N, H)
    assert np.array_equal(st[:, :sub], ost)
    voiced_equal = (orec[:, 0] == 0.0) == (rec[:sub, 0] == 0.0)
    assert np.all(voiced_equal), "voiced / unvoiced decision differs"
    assert np.all(np.abs(rec[:sub, 0] - orec[:, 0]) <= 1e-4 * np.abs(orec[:, 0]))
    assert np.all(np.abs(rec[:sub, 1] - orec[:, 1]) <= 1e-4)
    assert np.all(np.abs(rec[:sub, 2:10:2] - orec[:, 2:10:2]) <= 1e-4 * np.abs(orec[:, 2:10:2]))
    for t in range(sub):
        assert np.all(rel_close(rec[t, 10:23], orec[t, 10:23])), t
        assert np.all(rel_close(rec[t, 23:36], orec[t, 23:36])), t


@pytest.mark.parametrize("order", [8, 10, 16, 20])
def test_analyze_frames_other_lpc_orders(vb, pkg, oracle, audio_d, order):
    """The fused kernel's register Levinson is built for order 12; another order runs from the LPC kernels beside the fused
    pitch + MFCC pass: same pitch and MFCC columns as the default call, LPC columns equal to the stand-alone entry point and
    within 1e-6 of the oracle."""
    F = 96
    base, _ = vb.analyze_frames(audio_d, pkg.AnalysisParams.make(SR), frame_len=N, stride=H, n_frames=F)
    params = pkg.AnalysisParams.make(SR, lpc_order=order)
    cols = params.columns()
    rec, st = vb.analyze_frames(audio_d, params, frame_len=N, stride=H, n_frames=F)
    assert rec.shape == (F, cols["lpc"][0] + order + 1) and np.all(st == 0)
    assert np.array_equal(rec[:, 0:2], base[:, 0:2]) and np.array_equal(rec[:, 2:23], base[:, 2:23])      # pitch, formants, MFCC
    han = vb.window(pkg.WINDOW_HANNING, N)
    _, a = vb.autocorr_lpc(audio_d, order, frame_len=N, stride=H, n_frames=F, window=han)
    l0 = cols["lpc"][0]
    assert np.array_equal(rec[:, l0:l0 + order + 1], a)
    audio = audio_d.numpy()
    w = oracle.window("hanning", N)
    for t in range(0, F, 7):
        exp = oracle.lpc(oracle.autocorrelate(audio[t * H:t * H + N] * w, order + 1), order)
        assert np.all(rel_close(rec[t, l0:l0 + order + 1], exp)), t


def test_analyze_frames_parts_can_be_skipped_and_rows_can_be_padded(vb, pkg, audio_d):
    F = 64
    full = pkg.AnalysisParams.make(SR)
    rec, _ = vb.analyze_frames(audio_d, full, frame_len=N, stride=H, n_frames=F)
    only_pitch = pkg.AnalysisParams.make(SR, lpc_order=0, formant_order=0, mfcc=None)
    assert int(vb.L.vbx_record_doubles(only_pitch)) == 2
    r2, st2 = vb.analyze_frames(audio_d, only_pitch, frame_len=N, stride=H, n_frames=F)
    assert r2.shape == (F, 2) and np.all(st2 == 0)
    assert np.all(np.abs(r2[:, 0] - rec[:, 0]) <= 1e-4 * np.abs(rec[:, 0]))
    no_mfcc = pkg.AnalysisParams.make(SR, mfcc=None)
    assert no_mfcc.columns() == {"pitch": (0, 2), "formants": (2, 8), "lpc": (10, 13)}
    r3, _ = vb.analyze_frames(audio_d, no_mfcc, frame_len=N, stride=H, n_frames=F, record_ld=40)
    assert r3.shape == (F, 40)
    assert np.array_equal(r3[:, 2:10], rec[:, 2:10]) and np.array_equal(r3[:, 10:23], rec[:, 23:36])
    with pytest.raises(pkg.VoxBoxError):          # rows too short / odd
        vb.analyze_frames(audio_d, full, frame_len=N, stride=H, n_frames=F, record_ld=35)
    # empty batch
    r0, _ = vb.analyze_frames(audio_d, full, frame_len=N, stride=H, n_frames=0)
    assert r0.shape == (0, 36)


@pytest.mark.parametrize("N,H,sr", [(1103, 441, 44100.0), (1102, 441, 44100.0), (1025, 512, 48000.0), (1199, 480, 48000.0), (882, 441, 44100.0),
                                    (1201, 600, 48000.0), (2047, 1024, 48000.0), (2049, 1024, 48000.0), (4095, 2048, 48000.0)])
def test_mfcc_inside_the_fused_kernel_at_lengths_that_do_not_divide_the_transform(pkg, monkeypatch, N, H, sr):
    """MFCC::mfcc (src/spectrum.rs:401-441) of a frame whose length does not divide the fused kernel's transform: the bins are
    interpolated from the transform's (mfcc_interp_t) instead of coming from a chirp-z kernel beside it.  Same records: the MFCC
    columns within 1e-11 of the chirp-z form's (VBX_MFCC_INTERP=0: the tables' design error is < 1e-14 of the largest bin,
    tests/test_mfcc_interp_table.py), every other column and status bit for bit; and the form is really the one that ran."""
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VBX_MFCC_INTERP", mode)
        ctx = pkg.VoxBox(0)
        try:
            audio = ctx.synth_speech(int(8 * sr), sample_offset=int(2 * sr))
            F = pkg.frame_count(int(8 * sr), N, H)
            est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
            params = pkg.AnalysisParams.make(sr, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0, mfcc=(13, 100.0, 8000.0))
            ctx.profile_reset(); ctx.profile(True)
            rec, st = ctx.analyze_frames(audio, params, frame_len=N, stride=H, n_frames=F)
            names = set(ctx.profile_streams().keys())
            ctx.profile(False)
            got[mode] = (rec.copy(), st.copy(), names)
            audio.free()
        finally:
            ctx.close()
    (r0, s0, n0), (r1, s1, n1) = got["0"], got["1"]
    # ("mfcc_rows" = log10 + DCT of the fused call's filter sums, a lane per row since round 6: not a transform)
    assert any(k.startswith("mfcc") and k != "mfcc_rows" for k in n0), n0         # the chirp-z kernel beside the fused one
    assert not any(k.startswith("mfcc") and k != "mfcc_rows" for k in n1), n1     # no MFCC kernel at all: the bins came from the fused kernel
    assert np.array_equal(s0, s1)
    assert np.array_equal(r0[:, :10], r1[:, :10]) and np.array_equal(r0[:, 23:], r1[:, 23:])
    assert np.abs(r0[:, 10:23] - r1[:, 10:23]).max() <= 1e-11           # (observed over 720,000 frames: 6e-14)
    assert np.abs(r0[:, 10:23]).max() > 5.0


@pytest.mark.parametrize("N,H,band", [(1200, 480, (0.0, 20000.0)), (1200, 480, (100.0, 8000.0)), (1024, 512, (50.0, 22000.0)), (800, 320, (0.0, 16000.0)),
                                      (2048, 1024, (0.0, 20000.0)), (600, 240, (300.0, 23000.0))])
def test_fused_mfcc_with_filters_above_a_quarter_of_the_rate(vb, pkg, oracle, audio_d, N, H, band):
    """The fused kernels' MFCC products take the bins of P[m] alone when no mirrored bin n/2 - m/q can be one of the filters' (the
    speech settings: round 5's one-sided form) and both sides otherwise: filters that reach up to 20-23 kHz of 24 against the
    oracle, 40 filters among them."""
    audio = audio_d.numpy()
    F = min(pkg.frame_count(audio.size, N, H), 60)
    w = oracle.window("hanning", N)
    for k in (13, 40):
        params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=0, mfcc=(k, band[0], band[1]))
        rec, st = vb.analyze_frames(audio_d, params, frame_len=N, stride=H, n_frames=F)
        for t in range(0, F, 4):
            s, m = oracle.mfcc(audio[t * H:t * H + N] * w, k, band[0], band[1], SR)
            assert st[2, t] == s
            if s == 0:
                assert np.all(rel_close(rec[t, 2:2 + k], m)), (k, t, float(np.abs(rec[t, 2:2 + k] - m).max()))


@pytest.mark.parametrize("N,H,sr", [(1103, 441, 44100.0), (1199, 480, 48000.0), (1600, 640, 48000.0), (3000, 1200, 48000.0)])
def test_interpolated_mfcc_of_a_pure_tone(vb, pkg, oracle, N, H, sr):
    """The frame with the largest dynamic range a spectrum can have: one sinusoid under the Hanning window, no noise floor -- most mel
    filters hold nothing but its leakage, 1e-8 .. 1e-12 of the peak.  The interpolated bins' error is a fraction of the LARGEST bin
    (< 1e-14: tests/test_mfcc_interp_table.py), so these filters are where it would show: every coefficient within 1e-6 of the oracle's."""
    t = np.arange(int(0.5 * sr))
    for f0 in (440.0, 1000.0, 3217.3):
        x = 0.5 * np.sin(2 * np.pi * f0 * t / sr)
        F = pkg.frame_count(x.size, N, H)
        params = pkg.AnalysisParams.make(sr, pitch=(0.2, 75.0, 600.0), lpc_order=0, formant_order=0, mfcc=(13, 100.0, 8000.0))
        rec, st = vb.analyze_frames(x, params, frame_len=N, stride=H, n_frames=F)
        w = oracle.window("hanning", N)
        for i in range(0, F, 5):
            s, m = oracle.mfcc(x[i * H:i * H + N] * w, 13, 100.0, 8000.0, sr)
            assert st[2, i] == s == 0
            assert np.all(rel_close(rec[i, 2:15], m)), (f0, i, float(np.abs(rec[i, 2:15] - m).max()))


def test_interpolated_mfcc_at_random_shapes(pkg, monkeypatch):
    """Thirty seeded random (frame length, hop, sample rate, filter count, band) combinations through the fused call and through
    vbx_mfcc_f64, with and without the interpolated form: the same statuses, MFCC within 1e-11 (relative to the row's largest
    coefficient, floor 1), every other column bit for bit.  Odd and even lengths and hops, bands from 0 Hz, up to 40 filters,
    all four transform plans."""
    rng = np.random.default_rng(20250105)
    cases = []
    while len(cases) < 30:
        n = int(rng.integers(520, 4090))
        sr = float(rng.choice([16000.0, 22050.0, 32000.0, 44100.0, 48000.0]))
        hop = int(rng.integers(n // 4, n))
        k = int(rng.choice([8, 13, 13, 20, 26, 40]))
        lo = float(rng.choice([0.0, 50.0, 100.0, 300.0]))
        hi = float(rng.choice([0.12, 0.17, 0.22, 0.22, 0.4]) * sr)      # (above a quarter of the rate there is no interpolated form: the other kernels)
        bins, bad = pkg.mfcc_bins(n, k, lo, hi, sr)
        if bad or bins[-1] - bins[0] < 2:
            continue
        cases.append((n, hop, sr, k, lo, hi))
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VBX_MFCC_INTERP", mode)
        ctx = pkg.VoxBox(0)
        try:
            out = []
            for (n, hop, sr, k, lo, hi) in cases:
                ns = int(1.5 * sr) + 2 * n
                audio = ctx.synth_speech(ns, sample_offset=int(3 * sr))
                F = pkg.frame_count(ns, n, hop)
                params = pkg.AnalysisParams.make(sr, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=0, mfcc=(k, lo, hi))
                rec, st = ctx.analyze_frames(audio, params, frame_len=n, stride=hop, n_frames=F)
                han = ctx.window(pkg.WINDOW_HANNING, n)
                mf, mst = ctx.mfcc(audio, k, (lo, hi), sr, frame_len=n, stride=hop, n_frames=F, window=han)
                out.append((rec.copy(), st.copy(), np.array(mf).copy(), np.array(mst).copy()))
                audio.free()
            res[mode] = out
        finally:
            ctx.close()
    used = 0
    for case, a, b in zip(cases, res["0"], res["1"]):
        k = case[3]
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]), case
        assert np.array_equal(a[0][:, :2], b[0][:, :2]) and np.array_equal(a[0][:, 2 + k:2 + k + P + 1], b[0][:, 2 + k:2 + k + P + 1]), case   # (a padding column may follow)
        for x, y in ((a[0][:, 2:2 + k], b[0][:, 2:2 + k]), (a[2], b[2])):
            scale = np.maximum(np.abs(x).max(axis=1, keepdims=True), 1.0)
            assert np.all(np.abs(x - y) <= 1e-11 * scale), (case, float(np.abs(x - y).max()))
        used += int(not np.array_equal(a[0][:, 2:2 + k], b[0][:, 2:2 + k]))
    assert used >= 8, used                                                # a third of them really took the interpolated form (the rest: bins above a quarter of the transform, or too many for the LDS budget)


@pytest.mark.parametrize("N,H,sr,interp", [(1103, 441, 44100.0, 1), (1600, 640, 48000.0, 1), (2047, 1024, 48000.0, 1), (3000, 1200, 48000.0, 1), (4000, 2000, 48000.0, 1),
                                           (882, 441, 44100.0, 0), (1280, 640, 48000.0, 0)])
def test_mfcc_alone_by_one_transform_and_interpolated_bins(pkg, oracle, monkeypatch, N, H, sr, interp):
    """vbx_mfcc_f64 (MFCC::mfcc, src/spectrum.rs:401-441) at a length that does not divide a transform: the forward transform of the
    zero-padded frame + interpolated bins where that is the faster form (no matrix-core plan, or >= 1400 samples), the kernels of
    rounds 1-4 elsewhere and under VBX_MFCC_INTERP=0 -- within 1e-11 of each other, 1e-6 of the oracle."""
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VBX_MFCC_INTERP", mode)
        ctx = pkg.VoxBox(0)
        try:
            audio = ctx.synth_speech(int(6 * sr), sample_offset=int(2 * sr))
            F = pkg.frame_count(int(6 * sr), N, H)
            han = ctx.window(pkg.WINDOW_HANNING, N)
            mf, st = ctx.mfcc(audio, 13, (100.0, 8000.0), sr, frame_len=N, stride=H, n_frames=F, window=han)
            got[mode] = (np.array(mf).copy(), np.array(st).copy(), int(ctx.L.vbx_internal_last_mfcc_interp(ctx.ctx)), audio.numpy().copy())
            audio.free()
        finally:
            ctx.close()
    assert got["0"][2] == 0 and got["1"][2] == interp
    assert np.array_equal(got["0"][1], got["1"][1]) and not got["1"][1].any()
    assert np.abs(got["0"][0] - got["1"][0]).max() <= 1e-11
    x, w = got["1"][3], oracle.window("hanning", N)
    for t in range(0, 40, 3):
        s, m = oracle.mfcc(x[t * H:t * H + N] * w, 13, 100.0, 8000.0, sr)
        assert s == 0 and np.all(rel_close(got["1"][0][t], m)), t


@pytest.mark.parametrize("N,H", [(3000, 1200), (2500, 1000), (4096, 2048), (4000, 2000), (4095, 2048), (2205, 882)])
def test_the_4096_point_plan_as_two_kernels(pkg, monkeypatch, N, H):
    """Frames of 2049..4096 samples: the transforms + LPC + MFCC in one kernel, the lag curve through a scratch row, then the peak scan,
    the refinement (eleven frames per CU) and the far frames in kernels of their own (SP_ANALYZE_SPLIT: the default) -- bit for bit
    the fused kernel's records, candidates, counts and statuses (VBX_POW2_SPLIT=0).  Odd lengths too: their whole curve goes to the scratch
    row (with the fused kernel's exact last lags) for the far frames, the scan and the refinement read no more of it than of an even one."""
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VBX_POW2_SPLIT", mode)
        ctx = pkg.VoxBox(0)
        try:
            audio = ctx.synth_speech(20 * 48000, sample_offset=2 * 48000)
            F = pkg.frame_count(20 * 48000, N, H)
            est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
            params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0, mfcc=(13, 100.0, 8000.0))
            rec, st = ctx.analyze_frames(audio, params, frame_len=N, stride=H, n_frames=F)
            split_a = int(ctx.L.vbx_internal_last_spectral_split(ctx.ctx))
            han = ctx.window(pkg.WINDOW_HANNING, N)
            c1 = ctx.pitch(audio, SR, 0.2, 75.0, 600.0, kmax=1, frame_len=N, stride=H, n_frames=F, window=han)
            c8 = ctx.pitch(audio, SR, 0.2, 75.0, 600.0, kmax=8, frame_len=N, stride=H, n_frames=F, window=han)
            split_p = int(ctx.L.vbx_internal_last_spectral_split(ctx.ctx))
            got[mode] = ([rec.copy(), st.copy()] + [np.array(x).copy() for x in c1] + [np.array(x).copy() for x in c8], split_a, split_p)
            audio.free()
        finally:
            ctx.close()
    assert got["0"][1:] == (0, 0) and got["1"][1:] == (1, 1), (got["0"][1:], got["1"][1:])
    for a, b in zip(got["0"][0], got["1"][0]):
        assert np.array_equal(a, b)
    assert np.count_nonzero(got["1"][0][0][:, 0]) > 50                      # voiced frames among them


def test_profile_says_which_stream_a_kernel_ran_on(vb, pkg, audio_d):
    """vbx_profile_stream (ABI 5): the fused call's spectral kernel runs on the context's stream (0), the formant chain beside
    it on the side stream (1); a kernel that was not profiled reports -1.  bench.py ranks its dominant kernel among stream 0."""
    import ctypes as C
    F = pkg.frame_count(6 * 48000, N, H)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0, mfcc=(13, 100.0, 8000.0))
    vb.profile_reset(); vb.profile(True)
    vb.analyze_frames(audio_d, params, frame_len=N, stride=H, n_frames=F)
    streams = vb.profile_streams()
    vb.profile(False)
    assert streams["analyze"] == 0, streams
    assert streams["burg_lags"] == 1 and streams["formant_resonances"] == 1, streams
    sid = C.c_int(7)
    assert vb.L.vbx_profile_stream(vb.ctx, b"no_such_kernel", C.byref(sid)) == 0 and sid.value == -1


def test_rccl_record_gather_single_rank(vb, pkg):
    """The library's RCCL path on one GPU: communicator of world 1, loopback ncclSend/ncclRecv self-test, and the
    gather entry point (own rows in place / copied, slots, device-side wait)."""
    comm = pkg.Comm(vb, pkg.comm_unique_id(), 1, 0)
    try:
        comm.selftest(1 << 16)
        rows, rec = 1000, 36
        src = np.arange(rows * rec, dtype=np.float64).reshape(rows, rec)
        d_src = vb.to_device(src)
        d_out = vb.zeros((rows, rec))
        comm.gather_records(d_src, [rows], rec, dst=0, out=d_out, slot=1)       # copy path
        comm.wait(1)
        comm.sync()
        assert np.array_equal(d_out.numpy(), src)
        comm.gather_records(d_out, [rows], rec, dst=0, out=d_out, slot=2)       # in place: nothing moves
        comm.sync()
        assert np.array_equal(d_out.numpy(), src)
        with pytest.raises(pkg.VoxBoxError):
            comm.gather_records(d_src, [rows], rec, dst=1, out=d_out, slot=0)   # dst outside the communicator
        with pytest.raises(pkg.VoxBoxError):
            comm.gather_records(d_src, [rows], rec, dst=0, out=d_out, slot=9)
        d_src.free(); d_out.free()
    finally:
        comm.close()


def test_rows_kernel_forms_are_bit_identical_to_the_in_wavefront_forms(pkg, monkeypatch):
    """Round 6 moved Levinson and MFCC's log10 + DCT out of the 1200-point kernel's wavefront into a lane-per-record kernel (same
    operations in the same order).  VBX_MFCC_DEFER=0 keeps the MFCC tail in the wavefront: every record column bit for bit the same;
    and with VBX_LPC_EXACT=0 (no probe, no double-double redo) the LPC column too is what one f64 recursion on the same lag sums gives --
    checked against vbx_lpc_f64 on the autocorrelation rows vbx_autocorrelate_f64 returns for the same frames (the same two transforms)."""
    N, H, sr = 1200, 480, 48000.0
    got = {}
    for defer, exact in (("1", "0"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("VBX_MFCC_DEFER", defer)
        monkeypatch.setenv("VBX_LPC_EXACT", exact)
        with pkg.VoxBox(0) as ctx:
            audio = ctx.synth_speech(int(20 * sr), sample_offset=int(3 * sr))
            F = pkg.frame_count(int(20 * sr), N, H)
            params = pkg.AnalysisParams.make(sr, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=0, mfcc=(13, 100.0, 8000.0))
            rec, st = ctx.analyze_frames(audio, params, frame_len=N, stride=H, n_frames=F)
            if (defer, exact) == ("1", "0"):
                han = ctx.window(pkg.WINDOW_HANNING, N)
                r = ctx.autocorrelate(audio, N, frame_len=N, stride=H, n_frames=F, window=han)
                a_rows = ctx.lpc(r[:, :P + 1].copy(), P)
            got[(defer, exact)] = (rec.copy(), st.copy())
    base, st0 = got[("1", "0")]
    other, st1 = got[("0", "0")]
    assert np.array_equal(base.view(np.int64), other.view(np.int64)) and np.array_equal(st0, st1)
    l0, ln = params.columns()["lpc"]
    assert np.array_equal(base[:, l0:l0 + ln].view(np.int64), np.asarray(a_rows).view(np.int64))
    # the default build differs from it on the listed rows' LPC column only
    dflt, _ = got[("1", "1")]
    diff = np.any(dflt.view(np.int64) != base.view(np.int64), axis=1)
    cols = np.flatnonzero(np.any(dflt.view(np.int64) != base.view(np.int64), axis=0))
    assert diff.sum() <= F // 50 and (cols.size == 0 or (cols.min() >= l0 and cols.max() < l0 + ln)), (diff.sum(), cols)
