"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot walk
millions of frames): layout independence (hop-strided view == dense copy), batch/chunk independence,
run-to-run determinism, exact power-of-two scaling laws, plus random spot checks against the oracle
inside the big batch.  Needs a real MI355X."""
import numpy as np
import pytest

from conftest import rel_close

pytestmark = pytest.mark.gpu
SR, N, H, P = 48000.0, 1200, 480, 12


def test_config2_one_million_dense_frames(vb, oracle, pkg):
    """BASELINE config 2: 1M x 512 f64 frames, autocorrelate(13) -> lpc(12)."""
    F, n = 1_000_000, 512
    x = vb.synth_speech(F * n, sample_offset=12345)
    w = vb.window(pkg.WINDOW_HANNING, n)
    r, a = vb.empty((F, 13)), vb.empty((F, 13))
    vb.autocorr_lpc(x, P, frame_len=n, stride=n, n_frames=F, window=w, out=(r, a))
    R, A = r.numpy(), a.numpy()
    assert np.all(np.isfinite(R)) and np.all(np.isfinite(A)) and np.all(A[:, 0] == 1.0)
    assert np.all(R[:, 0] >= np.abs(R).max(axis=1) * (1 - 1e-12))       # windowed frames: r[0] is the maximum
    # determinism
    r2, a2 = vb.empty((F, 13)), vb.empty((F, 13))
    vb.autocorr_lpc(x, P, frame_len=n, stride=n, n_frames=F, window=w, out=(r2, a2))
    assert np.array_equal(R, r2.numpy()) and np.array_equal(A, a2.numpy())
    # chunk independence: a sub-batch that starts mid-way (different wave/frame alignment) gives the same bits
    lo, cnt = 333_337, 10_001
    xs = x.numpy()[lo * n:(lo + cnt) * n]
    rs, as_ = vb.autocorr_lpc(xs.reshape(cnt, n), P, window=w)
    assert np.array_equal(rs, R[lo:lo + cnt]) and np.array_equal(as_, A[lo:lo + cnt])
    # exact scaling law: x -> 4x gives r -> 16 r bit for bit, and the same LPC coefficients
    r4, a4 = vb.autocorr_lpc(4.0 * xs.reshape(cnt, n), P, window=w)
    assert np.array_equal(r4, 16.0 * rs) and np.array_equal(a4, as_)
    # spot checks against the oracle
    wh = oracle.window("hanning", n)
    xh = xs.reshape(cnt, n)
    for t in np.random.default_rng(0).integers(0, cnt, 40):
        er = oracle.autocorrelate(xh[t] * wh, 13)
        assert np.all(rel_close(rs[t], er)) and np.all(rel_close(as_[t], oracle.lpc(er, P)))
    for d in (x, r, a, r2, a2):
        d.free()


def test_pipeline_one_hour_strided(vb, oracle, pkg):
    """1 h of 48 kHz audio, 25 ms / 10 ms hop (359,998 frames): pitch + find_formants + MFCC."""
    ns = 3600 * 48000
    audio = vb.synth_speech(ns, sample_offset=7 * 48000)
    F = pkg.frame_count(ns, N, H)
    assert F == 359_998
    han = vb.window(pkg.WINDOW_HANNING, N)
    cand, cnt, st = vb.empty((F, 2, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32)
    vb.pitch(audio, SR, 0.2, 75.0, 600.0, kmax=2, frame_len=N, stride=H, n_frames=F, window=han, out=(cand, cnt, st))
    C, K, S = cand.numpy(), cnt.numpy(), st.numpy()
    assert np.all(S == 0) and np.all(K >= 1)
    top = C[:, 0, :]
    # the filter (src/periodic.rs:439) acts on the parabolic estimate; Brent may then move the lag by < 1
    assert np.all((top[:, 0] == 0.0) | ((top[:, 0] > SR / (SR / 75.0 + 1.0)) & (top[:, 0] < SR / (SR / 600.0 - 1.0))))
    assert np.all(top[:, 1] >= 0.2) and np.all(top[:, 1] <= 1.0)          # strength >= threshold; reflected at 1
    assert np.all(C[:, 0, 1] >= C[:, 1, 1])                                # sorted by descending strength
    sec = ((np.arange(F) * H + 7 * 48000) // 48000) % 5
    inner = (np.arange(F) * H + N + 7 * 48000) // 48000 % 5                 # frame entirely inside the second
    voiced = (sec != 4) & (inner != 4) & (sec == inner)
    unvoiced = (sec == 4) & (inner == 4)
    assert np.mean(top[voiced, 0] > 0) > 0.99 and np.mean(top[unvoiced, 0] == 0.0) > 0.99
    # the synthetic f0 glides 90..250 Hz: voiced estimates must sit on it (or its octave neighbours at most rarely)
    fv = top[voiced, 0]
    assert np.mean((fv > 85) & (fv < 260)) > 0.97
    # layout independence: dense copies of frames give bit-identical results to the strided view
    idx = np.sort(np.random.default_rng(1).integers(0, F, 300))
    ah = audio.numpy()
    dense = np.stack([ah[t * H:t * H + N] for t in idx])
    c2, k2, s2 = vb.pitch(dense, SR, 0.2, 75.0, 600.0, kmax=2, window=han)
    assert np.array_equal(c2, C[idx]) and np.array_equal(k2, K[idx])
    # exact top-k pruning at scale: the entries returned for kmax = 1 are, bit for bit, the head of the list returned for
    # kmax = 2 (one class of kmax: 1..3), and the kmax = 8 list the head of the kmax = 64 list (the other class: from 4 on
    # few-candidate frames refine four candidates at a time, which changes the last bits of the sinc sums) -- all 359,998
    # frames; between the classes the top candidate agrees within the Brent iteration's own scatter
    c1, k1, s1 = vb.empty((F, 1, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32)
    vb.pitch(audio, SR, 0.2, 75.0, 600.0, kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=(c1, k1, s1))
    c64, k64, s64 = vb.empty((F, 64, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32)
    vb.pitch(audio, SR, 0.2, 75.0, 600.0, kmax=64, frame_len=N, stride=H, n_frames=F, window=han, out=(c64, k64, s64))
    C64 = c64.numpy()
    assert np.array_equal(k1.numpy(), K) and np.array_equal(k64.numpy(), K)
    assert np.array_equal(c1.numpy()[:, 0], C[:, 0])
    c8, k8, s8 = vb.empty((F, 8, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32)
    vb.pitch(audio, SR, 0.2, 75.0, 600.0, kmax=8, frame_len=N, stride=H, n_frames=F, window=han, out=(c8, k8, s8))
    assert np.array_equal(c8.numpy(), C64[:, :8]) and np.array_equal(k8.numpy(), K)
    for d in (c8, k8, s8):
        d.free()
    t1, t64 = C[:, 0], C64[:, 0]
    far = ~((np.abs(t1[:, 0] - t64[:, 0]) <= 1e-6 * np.abs(t64[:, 0])) & (np.abs(t1[:, 1] - t64[:, 1]) <= 1e-5))
    # what is left: the two best candidates closer in strength than the scatter (they swap), at most a handful per 360,000
    assert far.sum() <= 8 and np.all(np.abs(C64[far, 0, 1] - C64[far, 1, 1]) <= 2e-5), (int(far.sum()), C[far][:3], C64[far, :2][:3])
    full = K <= 64                                                          # frames whose whole list fits
    assert full.sum() > 0.7 * F and np.all(np.diff(C64[full][:, :8, 1], axis=1) <= 0.0)
    for d in (c1, k1, s1, c64, k64, s64):
        d.free()
    # spot checks against the oracle inside the big batch (top candidate, BASELINE tolerance)
    wh = oracle.window("hanning", N)
    bad = 0
    for j in range(0, 300, 6):
        es, ec, en = oracle.pitch(dense[j] * wh, SR, 0.2, 75.0, 600.0)
        assert es == 0 and en == K[idx[j]]
        tie = en > 1 and abs(ec[0, 1] - ec[1, 1]) < 1e-3
        ok = abs(C[idx[j], 0, 0] - ec[0, 0]) <= 1e-4 * abs(ec[0, 0]) and abs(C[idx[j], 0, 1] - ec[0, 1]) <= 1e-4
        bad += 0 if (ok or tie) else 1
    assert bad == 0
    # find_formants + MFCC: finite, deterministic, chunk-independent at a segment boundary
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    seg = np.arange(0, F, 1000, dtype=np.int64)
    ff = vb.find_formants(audio, SR, P, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F, want=("formants", "status", "count"))
    assert np.all(ff["status"] == 0) and np.all(np.isfinite(ff["formants"]))
    assert np.all(ff["count"] >= 1) and np.all(ff["count"] <= 6)
    lo = 200_000
    sub = ah[lo * H:(lo + 999) * H + N]
    f2 = vb.find_formants(sub, SR, P, est0, frame_len=N, stride=H, want=("formants", "status"))
    assert np.array_equal(f2["formants"], ff["formants"][lo:lo + 1000])
    m, ms = vb.mfcc(audio, 13, (100.0, 8000.0), SR, frame_len=N, stride=H, n_frames=F, window=han)
    assert np.all(ms == 0) and np.all(np.isfinite(m))
    m2, _ = vb.mfcc(sub, 13, (100.0, 8000.0), SR, frame_len=N, stride=H, window=han)
    assert np.array_equal(m2, m[lo:lo + 1000])
    for t in (0, 417, 999):
        _, em = oracle.mfcc(sub[t * H:t * H + N] * wh, 13, 100.0, 8000.0, SR)
        assert np.all(rel_close(m2[t], em, 1e-6))
    audio.free()


def _pitch_flags(top, F, offset_samples):
    """Per-frame (voiced, unvoiced) expectation of the synthetic signal: every 5th second is noise only."""
    s0 = (np.arange(F) * H + offset_samples) // 48000 % 5
    s1 = (np.arange(F) * H + N + offset_samples) // 48000 % 5
    return (s0 != 4) & (s1 != 4) & (s0 == s1), (s0 == 4) & (s1 == 4)


def test_config3_ten_hours_pitch(vb, oracle, pkg):
    """BASELINE config 3 at its full size: 10 h of 48 kHz audio = 3,599,998 frames, PitchExtractor output per frame.
    Properties the oracle cannot walk: status / range / voicing statistics over every frame, run-to-run determinism,
    chunk independence (a sub-range started at an arbitrary frame gives the same bits), and 50 oracle spot checks."""
    ns = 10 * 3600 * 48000
    off = 3 * 48000
    audio = vb.synth_speech(ns, sample_offset=off)
    F = pkg.frame_count(ns, N, H)
    assert F == 3_599_998
    han = vb.window(pkg.WINDOW_HANNING, N)
    c1, k1, s1 = vb.empty((F, 1, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32)
    vb.pitch(audio, SR, 0.2, 75.0, 600.0, kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=(c1, k1, s1))
    C, K, S = c1.numpy()[:, 0, :], k1.numpy(), s1.numpy()
    assert np.all(S == 0) and np.all(K >= 1) and np.all(K <= pkg.pitch_max_candidates(N))
    assert np.all(C[:, 1] >= 0.2) and np.all(C[:, 1] <= 1.0)
    assert np.all((C[:, 0] == 0.0) | ((C[:, 0] > SR / (SR / 75.0 + 1.0)) & (C[:, 0] < SR / (SR / 600.0 - 1.0))))
    voiced, unvoiced = _pitch_flags(C, F, off)
    assert np.mean(C[voiced, 0] > 0) > 0.99 and np.mean(C[unvoiced, 0] == 0.0) > 0.99
    fv = C[voiced, 0]
    assert np.mean((fv > 85) & (fv < 260)) > 0.97
    # determinism
    c2, k2, s2 = vb.empty((F, 1, 2)), vb.empty(F, np.int32), vb.empty(F, np.int32)
    vb.pitch(audio, SR, 0.2, 75.0, 600.0, kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=(c2, k2, s2))
    assert np.array_equal(c2.numpy()[:, 0, :], C) and np.array_equal(k2.numpy(), K)
    # chunk independence: the same frames as their own batch, starting at an arbitrary frame of the recording
    lo, cnt = 1_234_567, 20_001
    c3, k3, s3 = vb.empty((cnt, 1, 2)), vb.empty(cnt, np.int32), vb.empty(cnt, np.int32)
    vb.pitch(audio.ptr + lo * H * 8, SR, 0.2, 75.0, 600.0, kmax=1, frame_len=N, stride=H, n_frames=cnt, window=han, out=(c3, k3, s3))
    assert np.array_equal(c3.numpy()[:, 0, :], C[lo:lo + cnt]) and np.array_equal(k3.numpy(), K[lo:lo + cnt])
    # oracle spot checks (top candidate within BASELINE tolerance, candidate count exact)
    wh = oracle.window("hanning", N)
    bad = swaps = 0
    for t in np.random.default_rng(3).integers(0, F, 50):
        fr = audio.numpy_slice(int(t) * H, N)
        es, ec, en = oracle.pitch(fr * wh, SR, 0.2, 75.0, 600.0)
        assert es == 0 and en == K[t]
        ok = abs(C[t, 0] - ec[0, 0]) <= 1e-4 * abs(ec[0, 0]) and abs(C[t, 1] - ec[0, 1]) <= 1e-4
        if not ok:
            near_tie = en > 1 and abs(ec[0, 1] - ec[1, 1]) < 1e-3 and abs(C[t, 0] - ec[1, 0]) <= 1e-4 * abs(ec[1, 0])
            swaps += int(near_tie)
            bad += int(not near_tie)
    assert bad == 0 and swaps <= 1
    for d in (audio, c1, k1, s1, c2, k2, s2, c3, k3, s3):
        d.free()


def test_config4_one_million_find_formants(vb, oracle, pkg):
    """BASELINE config 4 at its full size: 1M x 512 dense frames through find_formants (Burg 12 -> Laguerre roots ->
    resonances -> tracker, state reset every 1000 frames).  Determinism, chunk independence at a segment boundary,
    oracle spot checks of the per-frame resonances, and the tracker re-run on the CPU over whole segments (exact)."""
    F, n = 1_000_000, 512
    x = vb.synth_speech(F * n, sample_offset=777)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    seg = np.arange(0, F, 1000, dtype=np.int64)
    ff = vb.find_formants(x, SR, P, est0, seg_start=seg, frame_len=n, stride=n, n_frames=F)
    assert np.all(ff["status"] == 0) and np.all(np.isfinite(ff["formants"])) and np.all(np.isfinite(ff["coeffs"]))
    assert np.all(ff["count"] >= 0) and np.all(ff["count"] <= P // 2)
    f2 = vb.find_formants(x, SR, P, est0, seg_start=seg, frame_len=n, stride=n, n_frames=F, want=("formants", "status"))
    assert np.array_equal(f2["formants"], ff["formants"])                    # determinism
    # chunk independence: two whole utterances from the middle as their own batch
    lo = 500_000
    f3 = vb.find_formants(x.ptr + lo * n * 8, SR, P, est0, seg_start=np.array([0, 1000], dtype=np.int64), frame_len=n, stride=n,
                          n_frames=2000)
    assert np.array_equal(f3["formants"], ff["formants"][lo:lo + 2000]) and np.array_equal(f3["res"], ff["res"][lo:lo + 2000])
    # per-frame part against the oracle: Burg coefficients (1e-6) and resonance Hz (1e-4), count exact
    rng = np.random.default_rng(5)
    for t in rng.integers(0, F, 50):
        fr = x.numpy_slice(int(t) * n, n)
        es, _, eres, eco = oracle.find_formants(fr, SR, P, est0)
        assert es == 0
        assert np.all(rel_close(ff["coeffs"][t], eco))
        cnt = int(ff["count"][t])
        assert cnt == int(np.sum(eres[:, 0] != 0.0))
        assert np.all(np.abs(ff["res"][t, :cnt, 0] - eres[:cnt, 0]) <= 1e-4 * np.abs(eres[:cnt, 0]))
        assert np.all(ff["res"][t, cnt:] == 0.0)
    # the tracker over whole utterances, re-run by the oracle on the GPU's own resonance rows: exact
    for sgi in (0, 499, 999):
        est = est0.copy()
        for t in range(sgi * 1000, sgi * 1000 + 1000):
            est = oracle.estimate_formants(est, ff["res"][t])
            assert np.array_equal(est, ff["formants"][t]), (sgi, t)
    x.free()


def test_pipeline_shard_twelve_and_a_half_hours(vb, oracle, pkg):
    """The shard bench.py times at N = 1 (BASELINE config 5's per-GPU share: 12.5 h = 4,500,000 frames) through the
    fused call vbx_analyze_frames_f64: every record finite, statuses zero, determinism, the records of a sub-range
    (whole utterances) equal to their own batch, and oracle spot checks of all four record parts."""
    hours = 12.5
    F = int(hours * 3600 * 100)
    ns = (F - 1) * H + N
    off = 11 * 48000
    audio = vb.synth_speech(ns, sample_offset=off)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0, mfcc=(13, 100.0, 8000.0))
    REC = int(vb.L.vbx_record_doubles(params))
    seg = np.arange(0, F, 1000, dtype=np.int64)
    rec, st3 = vb.empty((F, REC)), vb.empty((3, F), np.int32)
    vb.analyze_frames(audio, params, seg_start=seg, frame_len=N, stride=H, n_frames=F, out=rec, record_ld=REC, status=st3)
    R, S3 = rec.numpy(), st3.numpy()
    assert np.all(S3 == 0) and np.all(np.isfinite(R))
    cols = {k: (c0, c0 + w) for k, (c0, w) in params.columns().items()}      # name -> [first, end) columns
    pit = R[:, cols["pitch"][0]:cols["pitch"][1]]
    assert np.all(pit[:, 1] >= 0.2) and np.all(pit[:, 1] <= 1.0)
    voiced, unvoiced = _pitch_flags(pit, F, off)
    assert np.mean(pit[voiced, 0] > 0) > 0.99 and np.mean(pit[unvoiced, 0] == 0.0) > 0.99
    assert np.all(R[:, cols["lpc"][0]] == 1.0)
    # determinism
    rec2 = vb.empty((F, REC))
    vb.analyze_frames(audio, params, seg_start=seg, frame_len=N, stride=H, n_frames=F, out=rec2, record_ld=REC, status=st3)
    assert np.array_equal(rec2.numpy(), R)
    rec2.free()
    # chunk independence: three utterances from the middle of the shard as their own batch
    lo = 2_345_000
    sub = vb.empty((3000, REC))
    vb.analyze_frames(audio.ptr + lo * H * 8, params, seg_start=np.array([0, 1000, 2000], dtype=np.int64), frame_len=N, stride=H,
                      n_frames=3000, out=sub, record_ld=REC, status=None)
    assert np.array_equal(sub.numpy(), R[lo:lo + 3000])
    sub.free()
    # oracle spot checks: pitch (1e-4), LPC (1e-6), MFCC (1e-6); formant tracks need the whole utterance: one of them
    wh = oracle.window("hanning", N)
    for t in np.random.default_rng(9).integers(0, F, 24):
        fr = audio.numpy_slice(int(t) * H, N)
        es, ec, en = oracle.pitch(fr * wh, SR, 0.2, 75.0, 600.0)
        ok = abs(pit[t, 0] - ec[0, 0]) <= 1e-4 * abs(ec[0, 0]) and abs(pit[t, 1] - ec[0, 1]) <= 1e-4
        assert es == 0 and (ok or (en > 1 and abs(ec[0, 1] - ec[1, 1]) < 1e-3)), t
        er = oracle.autocorrelate(fr * wh, P + 1)
        assert np.all(rel_close(R[t, cols["lpc"][0]:cols["lpc"][1]], oracle.lpc(er, P)))
        _, em = oracle.mfcc(fr * wh, 13, 100.0, 8000.0, SR)
        assert np.all(rel_close(R[t, cols["mfcc"][0]:cols["mfcc"][1]], em, 1e-6))
    sgi = 3210
    est = est0.copy()
    fo = R[:, cols["formants"][0]:cols["formants"][1]].reshape(F, 4, 2)
    for t in range(sgi * 1000, sgi * 1000 + 1000):
        fr = audio.numpy_slice(t * H, N)
        s_, est, _, _ = oracle.find_formants(fr, SR, P, est)
        assert s_ == 0 and np.all(np.abs(est[:, 0] - fo[t, :, 0]) <= 1e-4 * np.abs(est[:, 0])), t
    del R, S3, pit, fo
    # ---- the bench's DEFAULT mode: the whole shard is ONE utterance (bench.py --utterance-frames 0; what the reference's
    # loop over a file is, tests/lib.rs:75-79).  (1) the fused call's formant columns are vbx_find_formants_f64's tracks bit
    # for bit; (2) ALL 4,500,000 resonance rows of the GPU through the oracle's C estimate_formants, in order, on the host:
    # the tracks of the chunked scan (one utterance = 140,625 chunks of 32 frames, eight check + redo rounds, a sweep) are
    # the sequential scan's, bit for bit, on every frame of the configuration the driver times.
    vb.analyze_frames(audio, params, frame_len=N, stride=H, n_frames=F, out=rec, record_ld=REC, status=st3)
    R1 = rec.numpy()
    assert np.all(st3.numpy() == 0) and np.all(np.isfinite(R1))
    fo1 = np.ascontiguousarray(R1[:, cols["formants"][0]:cols["formants"][1]]).reshape(F, 4, 2)
    per_frame = [c for k, (a, b) in cols.items() if k != "formants" for c in range(a, b)]
    rec.free()
    ff = vb.find_formants(audio, SR, P, est0, frame_len=N, stride=H, n_frames=F, want=("formants", "res", "status"))
    assert np.all(ff["status"] == 0)
    assert np.array_equal(ff["formants"], fo1)
    trk = oracle.soak_track(ff["res"], ff["status"], est0)
    bad = np.flatnonzero(np.any(trk != ff["formants"], axis=(1, 2)))
    assert bad.size == 0, (bad.size, bad[:5])
    del ff, trk
    # the per-frame columns do not depend on how the recording is cut into utterances (the first launch's rows again)
    rec1k = vb.empty((F, REC))
    vb.analyze_frames(audio, params, seg_start=seg, frame_len=N, stride=H, n_frames=F, out=rec1k, record_ld=REC, status=st3)
    R2 = rec1k.numpy()
    assert np.array_equal(R2[:, per_frame], R1[:, per_frame])
    # ... and the tracks differ only after an utterance start of the 1,000-frame cut, never before the first one
    fo2 = R2[:, cols["formants"][0]:cols["formants"][1]].reshape(F, 4, 2)
    assert np.array_equal(fo2[:1000], fo1[:1000])
    for d in (audio, rec1k, st3):
        d.free()
