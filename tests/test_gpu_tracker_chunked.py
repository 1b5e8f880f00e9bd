"""The tracker scan of long utterances (k_tracker.hip: speculative chunks of 32 frames, warm-up, parallel repair rounds and a
final in-order sweep) must return the sequential scan's rows BIT FOR BIT, whatever the resonances look like: frames that
overwrite every estimate (the state is forgotten within a chunk: nothing to repair), frames with too few resonances (the
state persists: most chunks are redone), utterance boundaries anywhere relative to the chunk grid, skipped frames.
VBX_TRACKER_CHUNKED=1 / 0 selects the scan; without it utterances of 384 frames or more, and batches of 65,536 frames or more, take the
chunked one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR, N, H, P = 48000.0, 512, 512, 12


def _both(vb, monkeypatch, fn):
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VBX_TRACKER_CHUNKED", mode)
        out[mode] = fn()
    monkeypatch.delenv("VBX_TRACKER_CHUNKED")
    return out["0"], out["1"]


def _rows(rng, F, n_res, lo, hi, zero_rows=0.0):
    """F rows of n_res (frequency, bandwidth) entries: between lo and hi real resonances (ascending frequency), zero padded."""
    res = np.zeros((F, n_res, 2))
    k = rng.integers(lo, hi + 1, F)
    for t in range(F):
        if rng.random() < zero_rows:
            continue
        f = np.sort(rng.uniform(60.0, 8000.0, k[t]))
        res[t, :k[t], 0] = f
        res[t, :k[t], 1] = rng.uniform(20.0, 900.0, k[t])
    return res


@pytest.mark.parametrize("n_est", [1, 3, 4, 6])
@pytest.mark.parametrize("kind", ["rich", "sparse", "mixed"])
def test_chunked_scan_equals_sequential_scan(vb, pkg, monkeypatch, n_est, kind):
    rng = np.random.default_rng(100 * n_est + len(kind))
    F = 20000
    lo, hi, zr = {"rich": (4, 7, 0.0), "sparse": (0, 2, 0.3), "mixed": (0, 6, 0.1)}[kind]
    res = _rows(rng, F, 8, lo, hi, zr)
    if kind == "mixed":                                   # stretches of silence inside a long utterance
        for a in rng.integers(0, F - 400, 12):
            res[a:a + rng.integers(1, 300)] = 0.0
    est0 = np.array([[f, 60.0 + 10 * i] for i, f in enumerate(pkg.MALE_FORMANT_ESTIMATES)] + [[5000.0, 100.0], [6500.0, 120.0]])[:n_est]
    # utterances: one long stretch, many short ones, boundaries on / next to the chunk grid
    cuts = {0, 64, 65, 127, 128, 1000, 1001, 1063, 9000, 9001, 9002, 15000, 15064}
    cuts |= set(rng.integers(9100, 14000, 40).tolist())
    seg = np.array(sorted(cuts), dtype=np.int64)
    status = (rng.random(F) < 0.02).astype(np.int32) * 2
    dup = np.sort(np.concatenate([seg, seg[5:25], [F]]))                   # empty utterances among them, one that starts at F
    for segs, fs in ((None, None), (seg, None), (seg, status), (np.array([0], dtype=np.int64), status), (dup, status)):
        a, b = _both(vb, monkeypatch, lambda: vb.estimate_formants(res, est0, seg_start=segs, frame_status=fs))
        assert a.shape == (F, n_est, 2)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (kind, n_est, segs is None, fs is None,
                                                                      int(np.argmax(np.any(a != b, axis=(1, 2)))))


@pytest.mark.parametrize("n_est", [1, 2, 3, 4, 5, 6])
def test_index_form_equals_general_form(vb, pkg, oracle, monkeypatch, n_est):
    """The tracker's step on entry INDICES (rows as find_formants writes them: `count` real entries with ascending positive
    frequencies, then zeros) against the general step on (frequency, bandwidth) values (VBX_TRACKER_GENERAL=1): bit for bit,
    through both scans -- rows with 0..6 real entries, all-zero rows, skipped frames, estimates that start unsorted and
    on top of each other (duplicate picks, the unassigned-peak fill and its swaps), and rows that do NOT qualify mixed in
    (equal frequencies, a zero frequency among the real ones, more than six entries: the whole wavefront then takes the
    general step).  And against the oracle's sequential tracker on a stretch of the rows."""
    rng = np.random.default_rng(7 * n_est + 1)
    F = 12000
    res = _rows(rng, F, 8, 0, 6, 0.05)
    cnt = np.sum(res[:, :, 0] > 0, axis=1).astype(np.int32)
    res[:, :, 0] = np.round(res[:, :, 0] / 250.0) * 250.0 * (res[:, :, 0] > 0)      # coarse grid: near and exact ties of distance
    for t in range(F):                                                             # keep the rows strictly ascending
        k = cnt[t]
        f = np.unique(res[t, :k, 0][res[t, :k, 0] > 0])
        res[t, :, :] = 0.0
        res[t, :f.size, 0] = f
        res[t, :f.size, 1] = rng.uniform(20.0, 900.0, f.size)
        cnt[t] = f.size
    spoiled = res.copy(); scnt = cnt.copy()
    for t in rng.integers(0, F, 60):                                               # rows the index form must not take
        k = scnt[t]
        kind = rng.integers(0, 3)
        if kind == 0 and k >= 2: spoiled[t, 1, 0] = spoiled[t, 0, 0]               # equal frequencies
        elif kind == 1 and k >= 2: spoiled[t, 0, 0] = 0.0                          # a zero among the real entries
        else:
            spoiled[t, :7, 0] = np.sort(rng.uniform(100.0, 7000.0, 7)); spoiled[t, :7, 1] = 50.0; scnt[t] = 7
    est0 = np.array([[1000.0, 60.0], [1000.0, 70.0], [250.0, 80.0], [4000.0, 90.0], [3900.0, 100.0], [6500.0, 120.0]])[:n_est]
    seg = np.array(sorted({0, 100, 101, 5000, 5064, 9000} | set(rng.integers(9100, 11000, 20).tolist())), dtype=np.int64)
    status = (rng.random(F) < 0.03).astype(np.int32) * 2
    for rows, counts in ((res, cnt), (spoiled, scnt)):
        for chunked in ("0", "1"):
            monkeypatch.setenv("VBX_TRACKER_CHUNKED", chunked)
            monkeypatch.delenv("VBX_TRACKER_GENERAL", raising=False)
            a = vb.estimate_formants(rows, est0, seg_start=seg, frame_status=status, res_count=counts)
            monkeypatch.setenv("VBX_TRACKER_GENERAL", "1")
            b = vb.estimate_formants(rows, est0, seg_start=seg, frame_status=status, res_count=counts)
            monkeypatch.delenv("VBX_TRACKER_GENERAL")
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (n_est, chunked, int(np.argmax(np.any(a != b, axis=(1, 2)))))
        monkeypatch.delenv("VBX_TRACKER_CHUNKED")
    # the oracle's sequential tracker on the first utterances of the qualifying rows (zero-padded rows, as the reference passes them)
    a = vb.estimate_formants(res, est0, seg_start=seg, frame_status=status, res_count=cnt)
    est = est0.copy()
    for t in range(0, 400):
        if t in set(seg.tolist()):
            est = est0.copy()
        if status[t] == 0:
            est = oracle.estimate_formants(est, res[t])
        assert np.array_equal(a[t], est), (n_est, t)


def test_chunked_scan_short_batches_and_edges(vb, pkg, monkeypatch):
    """Batches shorter than a chunk or the warm-up, one-frame utterances, a batch that ends on the chunk grid."""
    rng = np.random.default_rng(7)
    est0 = np.array([[f, 80.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    for F in (1, 2, 63, 64, 65, 127, 128, 129, 192, 1000):
        res = _rows(rng, F, 6, 0, 6, 0.1)
        segs = [None, np.array([0], dtype=np.int64)]
        if F > 3:
            segs.append(np.arange(0, F, 1, dtype=np.int64))                     # every frame its own utterance
            segs.append(np.array(sorted({0, F // 2, F - 1}), dtype=np.int64))
            segs.append(np.array([0, 0, F // 3, F // 3, F // 3, F - 2, F - 2, F], dtype=np.int64))   # empty utterances, a start at F
        for s in segs:
            a, b = _both(vb, monkeypatch, lambda: vb.estimate_formants(res, est0, seg_start=s))
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (F, None if s is None else s.size)


def test_find_formants_long_utterance_takes_the_chunked_scan(vb, pkg, oracle, monkeypatch):
    """A single utterance of 6000 frames through vbx_find_formants_f64: by default the
    chunked scan runs (longest utterance >= 384 frames); its formant tracks equal the sequential scan's bit for bit and the oracle's frame loop within 1e-4."""
    F = 6000
    audio_d = vb.synth_speech(F * H + N, sample_offset=3 * 48000)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    try:
        default = vb.find_formants(audio_d, SR, P, est0, frame_len=N, stride=H, n_frames=F)
        seq, chk = _both(vb, monkeypatch, lambda: vb.find_formants(audio_d, SR, P, est0, frame_len=N, stride=H, n_frames=F))
        assert np.array_equal(seq["formants"].view(np.uint64), chk["formants"].view(np.uint64))
        assert np.array_equal(default["formants"].view(np.uint64), chk["formants"].view(np.uint64))
        audio = audio_d.numpy()
        est = est0.copy()
        for t in range(300):
            s, e, _, _ = oracle.find_formants(audio[t * H:t * H + N], SR, P, est)
            if s == 0:
                est = e
            assert s == default["status"][t]
            assert np.all(np.abs(default["formants"][t, :, 0] - est[:, 0]) <= 1e-4 * np.abs(est[:, 0])), t
    finally:
        audio_d.free()


def test_time_sliced_find_formants_equals_chunked(vb, pkg, monkeypatch):
    """VBX_TRACKER_CHUNKED=0 on a large batch of equal-length utterances takes round 2's first answer, the time-sliced
    find_formants (six slices, the sequential tracker on its own stream): same tracks, bit for bit, as the default path."""
    n_seg, seg_len = 160, 512
    F = n_seg * seg_len                                                   # 81,920 frames: above the slicing threshold
    audio_d = vb.synth_speech(F * 256 + N, sample_offset=48000)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    seg = np.arange(0, F, seg_len, dtype=np.int64)
    try:
        seq, chk = _both(vb, monkeypatch, lambda: vb.find_formants(audio_d, SR, P, est0, seg_start=seg, frame_len=N, stride=256, n_frames=F,
                                                               want=("formants", "status")))
        assert np.array_equal(seq["status"], chk["status"])
        assert np.array_equal(seq["formants"].view(np.uint64), chk["formants"].view(np.uint64))
        # the state really is reset at every utterance start: the first frame's estimates do not depend on the utterance before
        one = vb.find_formants(audio_d, SR, P, est0, frame_len=N, stride=256, n_frames=F, want=("formants",))
        assert not np.array_equal(one["formants"], chk["formants"])
    finally:
        audio_d.free()

