"""SURVEY 8f rows N2 / N3 on the GPU: 16-bit PCM ingestion + Windower framing, RMS::rms,
Filter::preemphasis -- against the oracle / the reference's own reader arithmetic."""
import os
import wave

import numpy as np
import pytest

from conftest import rel_close

pytestmark = pytest.mark.gpu


def _pcm(path):
    with wave.open(path, "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy(), float(w.getframerate())


def test_pcm16_ingestion_is_exact_and_feeds_find_formants(vb, oracle, pkg, golden_dir):
    """tests/lib.rs:15-19,44-90: WAV samples / 32767, rectangle Windower 1024/512, find_formants p=10 --
    the whole chain on the device from the raw int16 samples."""
    pcm, sr = _pcm(os.path.join(golden_dir, "short_sample.wav"))
    d = vb.pcm16_to_f64(pcm)
    host = pcm.astype(np.float64) / 32767.0
    assert np.array_equal(d.numpy(), host)                       # bit exact
    F = pkg.frame_count(pcm.size, 1024, 512)
    assert F == 4
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    out = vb.find_formants(d, sr, 10, est0, frame_len=1024, stride=512, n_frames=F)
    est = est0.copy()
    for t in range(F):
        st, est, _, _ = oracle.find_formants(host[t * 512:t * 512 + 1024], sr, 10, est)
        assert st == out["status"][t]
        assert np.all(np.abs(out["formants"][t] - est) <= 1e-4 * np.abs(est))
    every = np.arange(-32768, 32768, dtype=np.int16)             # exhaustive: the division must be exact everywhere
    assert np.array_equal(vb.pcm16_to_f64(every).numpy(), every.astype(np.float64) / 32767.0)
    assert np.array_equal(vb.pcm16_to_f64(every[3:-2]).numpy(), every[3:-2].astype(np.float64) / 32767.0)
    rng = np.random.default_rng(0)
    big = rng.integers(-32768, 32767, 1_000_003, dtype=np.int16)
    assert np.array_equal(vb.pcm16_to_f64(big).numpy(), big.astype(np.float64) / 32767.0)


@pytest.mark.parametrize("n", [64, 100, 512, 1200, 4096])
def test_rms(vb, oracle, n):
    """src/waves.rs:138-144 test_rms + random frames."""
    assert abs(vb.rms(oracle.sine(64, 64.0, 1.0)[None, :])[0] - 0.707) < 1e-3
    x = np.random.default_rng(n).uniform(-1, 1, (11, n))
    got = vb.rms(x)
    for f in range(11):
        assert abs(got[f] - oracle.rms(x[f])) <= 1e-13 * oracle.rms(x[f])


@pytest.mark.parametrize("n,factor", [(32, 0.1), (512, 0.1), (1200, 0.05), (100, 0.12), (4096, 0.02), (1200, -0.09)])
def test_preemphasis(vb, oracle, n, factor):
    """src/waves.rs:114-118 test_pe (sine(32), factor 0.1) + random frames; |2*pi*factor| < 1."""
    x = np.random.default_rng(n).uniform(-1, 1, (7, n))
    x[0] = oracle.sine(n, float(n), 1.0)
    got = vb.preemphasis(x, factor)
    for f in range(7):
        exp = oracle.preemphasis(x[f], factor)
        assert np.all(rel_close(got[f], exp, 1e-12, 1e-3)), (f, np.max(np.abs(got[f] - exp)))


@pytest.mark.parametrize("n,factor", [(1200, 0.25), (1200, 0.5), (4096, 0.25), (4096, 0.5), (512, -0.3), (4096, 1.0 / (2 * np.pi))])
def test_preemphasis_unstable_factors(vb, oracle, n, factor):
    """|2*pi*factor| >= 1: the recurrence grows like c^n (to inf at n = 4096); the reference's sequential values --
    finite, inf, and where it produces them NaN -- must come out position by position (no NaN where it has none)."""
    x = np.random.default_rng(n + 1).uniform(-1, 1, (5, n))
    x[1, n // 2:] = 0.0                                   # zeros meeting overflowing powers: the inf * 0 hazard
    x[2, :] = 0.0
    got = vb.preemphasis(x, factor)
    for f in range(5):
        exp = oracle.preemphasis(x[f], factor)
        assert np.array_equal(np.isnan(got[f]), np.isnan(exp)) and np.array_equal(np.isinf(got[f]), np.isinf(exp)), f
        fin = np.isfinite(exp)
        assert np.all(np.abs(got[f][fin] - exp[fin]) <= 1e-12 * np.abs(exp[fin]) + 1e-300), f


def test_preemphasis_strided_view(vb, oracle, pkg):
    audio = vb.synth_speech(48000, sample_offset=99)
    a = audio.numpy()
    F = pkg.frame_count(a.size, 1200, 480)
    got = vb.preemphasis(audio, 0.1, frame_len=1200, stride=480, n_frames=F)
    for t in (0, 1, F // 2, F - 1):
        assert np.all(rel_close(got[t], oracle.preemphasis(a[t * 480:t * 480 + 1200], 0.1), 1e-12, 1e-3))


@pytest.mark.parametrize("n,ratio", [(1200, 10000.0 / 48000.0), (1024, 0.5), (512, 11025.0 / 44100.0), (100, 2.0), (333, 0.37), (7, 1.7)])
def test_resample_linear_is_bit_identical_to_the_oracle(vb, oracle, n, ratio):
    """src/lib.rs:57-61 (sample 0.10 Linear + Converter; parity unpinned by the reference)."""
    x = np.random.default_rng(n).uniform(-1, 1, (5, n))
    got = vb.resample_linear(x, ratio)
    assert got.shape[1] == oracle.resampled_len(n, ratio)
    for f in range(5):
        assert np.array_equal(got[f], oracle.resample_linear(x[f], ratio)), f


def test_find_formants_with_resample_ratio(vb, oracle, pkg):
    """examples/formant_extraction usage: down-sample to 10 kHz, then find_formants at the new rate."""
    audio = vb.synth_speech(3 * 48000, sample_offset=48000)
    a = audio.numpy()
    N, H, ratio, sr2, p = 1200, 480, 10000.0 / 48000.0, 10000.0, 10
    F = pkg.frame_count(a.size, N, H)
    dense = vb.empty((F, int(oracle.resampled_len(N, ratio))))
    vb.resample_linear(audio, ratio, frame_len=N, stride=H, n_frames=F, out=dense)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    M = dense.shape[1]
    out = vb.find_formants(dense, sr2, p, est0, frame_len=M, stride=M, n_frames=F)
    est = est0.copy()
    bad = 0
    for t in range(F):
        st, est = oracle.find_formants_ratio(a[t * H:t * H + N], sr2, ratio, p, est)
        assert st == out["status"][t]
        bad += 0 if np.all(np.abs(out["formants"][t] - est) <= 1e-4 * np.abs(est)) else 1
    assert bad == 0


@pytest.mark.gpu
def test_ring_buffer_view_autocorrelate(vb, oracle):
    """`impl Autocorrelate for VecDeque` (src/periodic.rs:291-304): the deque's logical order, wrapped around the
    end of its ring, through vbx_ring_frames_f64 + vbx_autocorrelate_f64 == the oracle on the linearised deque."""
    rng = np.random.default_rng(5)
    cap, head, n, hop, F = 4096, 3000, 512, 160, 12          # frames cross the wrap point
    ring = rng.uniform(-1, 1, cap)
    logical = np.concatenate([ring[head:], ring[:head]])
    d = vb.ring_frames(ring, head, F, n, hop)
    dense = d.numpy()
    for t in range(F):
        assert np.array_equal(dense[t], logical[t * hop:t * hop + n])
    got = vb.autocorrelate(d, 13, frame_len=n, stride=n, n_frames=F)
    d.free()
    for t in range(F):
        exp = oracle.autocorrelate(logical[t * hop:t * hop + n], 13)
        assert np.all(rel_close(got[t], exp)), t
    with pytest.raises(Exception):
        vb.ring_frames(ring, head, 30, n, hop)               # 29*160 + 512 > 4096: longer than the deque can be


def _speech_pcm(vb, n_samples, offset=0):
    """The synthetic recording quantised to 16-bit PCM the way a WAV writer would (round to nearest, peak 0.9)."""
    a = vb.synth_speech(n_samples, sample_offset=offset)
    x = a.numpy()
    a.free()
    return np.clip(np.rint(x / np.max(np.abs(x)) * 0.9 * 32767.0), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("N,H,odd_start", [(1200, 480, 0), (1200, 480, 1), (1200, 481, 0), (1024, 512, 0), (1103, 441, 0), (512, 256, 0)])
def test_analyze_frames_pcm16_is_bit_identical_to_widening_first(vb, pkg, N, H, odd_start):
    """vbx_analyze_frames_pcm16 (the fused frame loop reading 16-bit PCM, 960 B of new samples per frame instead of 3840) ==
    vbx_pcm16_to_f64 followed by vbx_analyze_frames_f64, bit for bit: every record column and every status.  1200-sample
    frames take the PCM forms of the fused spectral kernel and of Burg (4-byte loads where the frame starts on a 4-byte
    boundary -- an odd first sample or an odd hop takes the 2-byte path); the other shapes are widened inside the library."""
    F = 700
    pcm = _speech_pcm(vb, (F - 1) * H + N + odd_start, offset=3 * 48000)[odd_start:]
    params = pkg.AnalysisParams.make(48000.0)
    seg = np.array([0, 250, 251], dtype=np.int64)
    d16 = vb.to_device(np.concatenate([np.zeros(odd_start, np.int16), pcm]), np.int16)
    ptr16 = d16.ptr + 2 * odd_start                              # a view that starts on an odd sample: 2-byte aligned only
    rec, st = vb.empty((F, 36)), vb.empty((3, F), np.int32)
    vb.analyze_frames_pcm16(ptr16, params, seg_start=seg, frame_len=N, stride=H, n_frames=F, out=rec, record_ld=36, status=st)
    wide = vb.pcm16_to_f64(pcm)
    assert np.array_equal(wide.numpy(), pcm.astype(np.float64) / 32767.0)
    rec2, st2 = vb.analyze_frames(wide, params, seg_start=seg, frame_len=N, stride=H, n_frames=F)
    R, S = rec.numpy(), st.numpy()
    assert np.array_equal(S, st2)
    bad = np.nonzero(np.any(R != rec2, axis=1))[0]
    assert bad.size == 0, (bad[:5], R[bad[0]], rec2[bad[0]])
    assert np.any(R[:, 0] > 0) and np.all(S == 0)                # voiced frames in there, every part ran
    for d in (d16, rec, st, wide):
        d.free()


def test_analyze_frames_pcm16_rectangular_seed_and_other_params(vb, pkg):
    """Parameter sets that leave the fused kernel's PCM form: LPC order 10 (its own kernel -> widened copy), no MFCC, no
    formants; and silence (Burg's Err(LPC) status) -- all bit-identical to the widen-first path."""
    N, H, F = 1200, 480, 300
    pcm = _speech_pcm(vb, (F - 1) * H + N, offset=9 * 48000)
    pcm[40 * H:43 * H + N] = 0                                   # a few all-zero frames
    wide = vb.pcm16_to_f64(pcm)
    for kw in (dict(lpc_order=10), dict(mfcc=None), dict(formant_order=0), dict(lpc_order=0, mfcc=(20, 50.0, 12000.0))):
        params = pkg.AnalysisParams.make(48000.0, **kw)
        a, sa = vb.analyze_frames_pcm16(pcm, params, frame_len=N, stride=H)
        b, sb = vb.analyze_frames(wide, params, frame_len=N, stride=H, n_frames=F)
        rec = int(vb.L.vbx_record_doubles(params))               # (an odd record is padded to an even row; the pad is never written)
        assert np.array_equal(sa, sb) and np.array_equal(a[:, :rec], b[:, :rec], equal_nan=True), kw   # silent frames: Levinson on zeros is NaN in both
    assert np.any(sb[1] == 1)                                    # the silent frames: Err(LPC) from Burg
    wide.free()
