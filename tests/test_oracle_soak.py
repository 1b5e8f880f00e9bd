"""The native soak walker (oracle/vbx_soak.c) adds no arithmetic of its own: on a small stretch its arrays equal, bit for
bit, what the per-frame oracle calls the other tests use return.  (CPU only; tests/test_gpu_soak.py holds the GPU to it.)"""
import importlib

import numpy as np

SR, P = 48000.0, 12


def test_soak_walker_equals_per_frame_oracle_calls(oracle, pkg):
    synth = importlib.import_module(pkg.__name__ + ".synth")
    N, H, first, count = 1200, 480, 3, 40
    audio = synth.synth_speech((first + count - 1) * H + N, sample_offset=4 * 48000 - 9000)   # the voiced -> unvoiced boundary at 4 s falls inside
    what = oracle.SOAK_PITCH | oracle.SOAK_LPC | oracle.SOAK_MFCC | oracle.SOAK_FORMANTS
    s = oracle.soak(audio, N, H, first, count, P, SR, what, n_threads=3)
    w = oracle.window("hanning", N)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    seg = np.array([0, 17], dtype=np.int64)
    trk = oracle.soak_track(s["res"], s["ff_status"], est0, seg)
    est = None
    for i in range(count):
        fr = audio[(first + i) * H:(first + i) * H + N]
        st, c, n = oracle.pitch(fr * w, SR, 0.2, 75.0, 600.0, cap=3)
        assert st == s["pitch_status"][i] and n == s["pitch_count"][i]
        assert np.array_equal(c[:min(n, 3)], s["pitch_top"][i, :min(n, 3)])
        r = oracle.autocorrelate(fr * w, P + 1)
        assert np.array_equal(r, s["r"][i]) and np.array_equal(oracle.lpc(r, P), s["a"][i])
        ms, m = oracle.mfcc(fr * w, 13, 100.0, 8000.0, SR)
        assert ms == s["mfcc_status"][i] and np.array_equal(m, s["mfcc"][i])
        if i in seg:
            est = est0.copy()
        fs, est, res, co = oracle.find_formants(fr, SR, P, est)
        assert fs == s["ff_status"][i] and np.array_equal(co, s["burg"][i]) and np.array_equal(res, s["res"][i])
        assert s["res_count"][i] == int(np.sum(res[:, 0] != 0.0))
        assert np.array_equal(est, trk[i]), i
    assert np.any(s["pitch_top"][:, 0, 0] > 0) and np.any(s["pitch_top"][:, 0, 0] == 0)   # voiced and unvoiced frames both seen


def test_soak_walker_dense_frames_and_failures(oracle):
    """Dense [F, 512] batches (config 2 / 4 shape) and an all-zero frame: Burg's Err(LPC) leaves the tracker state alone."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal(6 * 512)
    x[2 * 512:3 * 512] = 0.0
    s = oracle.soak(x, 512, 512, 0, 6, P, SR, oracle.SOAK_LPC | oracle.SOAK_FORMANTS, n_threads=2)
    assert s["ff_status"].tolist() == [0, 0, 1, 0, 0, 0] and s["res_count"][2] == 0 and np.all(s["res"][2] == 0)
    est0 = np.array([[320.0, 1.0], [1440.0, 1.0], [2760.0, 1.0], [3200.0, 1.0]])
    trk = oracle.soak_track(s["res"], s["ff_status"], est0)
    assert np.array_equal(trk[2], trk[1])
    est = est0.copy()
    for i in range(6):
        fs, est, _, _ = oracle.find_formants(x[i * 512:(i + 1) * 512], SR, P, est)
        assert np.array_equal(est, trk[i])
