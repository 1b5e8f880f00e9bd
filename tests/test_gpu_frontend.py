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


def test_preemphasis_strided_view(vb, oracle, pkg):
    audio = vb.synth_speech(48000, sample_offset=99)
    a = audio.numpy()
    F = pkg.frame_count(a.size, 1200, 480)
    got = vb.preemphasis(audio, 0.1, frame_len=1200, stride=480, n_frames=F)
    for t in (0, 1, F // 2, F - 1):
        assert np.all(rel_close(got[t], oracle.preemphasis(a[t * 480:t * 480 + 1200], 0.1), 1e-12, 1e-3))
