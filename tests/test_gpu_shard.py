"""The tracker across shard boundaries on the GPU (SURVEY 8e "Exception"), through the C ABI:

* one device playing every rank in turn: each shard of ONE utterance analysed with its warm-up frames, stitched to the
  previous shard's last row (vbx_track_stitch_f64), concatenated -- bit-identical to the single call over the whole
  recording, for 2, 3 and 8 shards, with a right guess (nothing rewritten) and with a wrong one (the repair step runs);
* 2, 4 and 8 rank PROCESSES, one GPU each, over RCCL: analyse, vbx_comm_stitch_tracks_f64, vbx_gather_records_f64 -- rank 0's
  gathered array against a single-rank run, bit for bit.  Each size is skipped (not passed) on a box with fewer GPUs.

Needs a real MI355X: run with `-m gpu`.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, H, SR, P = 1200, 480, 48000.0, 12


def _params(pkg):
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    return pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0, mfcc=(13, 100.0, 8000.0))


@pytest.mark.parametrize("world,seg_list", [(2, None), (3, None), (8, None), (3, [0, 2500, 2530, 9000])])
def test_shards_of_one_utterance_stitch_to_the_single_scan(vb, pkg, world, seg_list):
    F = 12_000
    audio = vb.synth_speech((F - 1) * H + N, sample_offset=7 * 48000)
    params = _params(pkg)
    REC = int(vb.L.vbx_record_doubles(params))
    seg = None if seg_list is None else np.array(seg_list, dtype=np.int64)
    whole, st_whole = vb.analyze_frames(audio, params, seg_start=seg, frame_len=N, stride=H, n_frames=F)
    assert np.all(st_whole == 0)
    got = np.zeros_like(whole)
    changed = vb.empty(1, np.int32)
    prev = None                                            # (records buffer, rows) of the previous shard
    total_changed = 0
    for r in range(world):
        pl = pkg.shard_plan(F, world, r, seg)
        first = pl.lo - pl.warm
        n = pl.hi - first
        lseg = pkg.shard_local_segments(pl, seg)
        rec = vb.empty((n, REC))
        st3 = vb.empty((3, n), np.int32)
        vb.analyze_frames(audio.ptr + first * H * 8, params, seg_start=lseg, frame_len=N, stride=H, n_frames=n, out=rec, record_ld=REC, status=st3)
        if pl.continues_prev:
            assert prev is not None and r > 0
            state = prev[0].ptr + ((prev[1] - 1) * REC + 2) * 8      # the previous shard's last formant row
            vb.track_stitch(rec.ptr + 2 * 8, n, REC, pl.warm, pl.stop, state, changed)
            total_changed += int(changed.numpy()[0])
        got[pl.lo:pl.hi] = rec.numpy()[pl.warm:]
        if prev is not None:
            prev[0].free()
        prev = (rec, n)
        st3.free()
    prev[0].free()
    changed.free(); audio.free()
    assert np.array_equal(got, whole)                      # every column of every record, bit for bit
    # 64 warm-up frames: the guess is right almost every time; where it is not, the stitch rewrote a few rows (observed: 0 at
    # 2 and 3 shards of this recording, 9 rows in all at 8) -- the result above is exact either way
    assert total_changed <= 16 * world


@pytest.mark.parametrize("warm", [0, 1, 5])
def test_stitch_repairs_a_wrong_guess(vb, pkg, oracle, warm):
    """A warm-up too short to forget the initial estimates: the shard's first rows are wrong until the stitch redoes them."""
    F, cut = 3000, 1500
    audio = vb.synth_speech((F - 1) * H + N, sample_offset=11 * 48000)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    whole = vb.find_formants(audio, SR, P, est0, frame_len=N, stride=H, n_frames=F, want=("formants", "status"))
    a = {"formants": vb.empty((cut, 4, 2)), "status": vb.empty(cut, np.int32)}
    vb.find_formants(audio, SR, P, est0, frame_len=N, stride=H, n_frames=cut, out=a)
    first = cut - warm
    n = F - first
    b = {"formants": vb.empty((n, 4, 2)), "status": vb.empty(n, np.int32)}
    vb.find_formants(audio.ptr + first * H * 8, SR, P, est0, frame_len=N, stride=H, n_frames=n, out=b)
    before = b["formants"].numpy()
    changed = vb.empty(1, np.int32)
    vb.track_stitch(b["formants"], n, 8, warm, n, a["formants"].ptr + (cut - 1) * 8 * 8, changed)
    after = b["formants"].numpy()
    nch = int(changed.numpy()[0])
    assert np.array_equal(after[warm:], whole["formants"][cut:])
    assert not np.array_equal(before[warm:], whole["formants"][cut:]) and nch > 0      # the guess WAS wrong, rows were rewritten
    assert nch < 200                                                                       # ... and only until the scans met
    assert np.array_equal(before[warm + nch:], after[warm + nch:])
    # a stitch on rows that are not the last call's is refused
    with pytest.raises(pkg.VoxBoxError):
        vb.track_stitch(a["formants"], cut, 8, 0, cut, a["formants"].ptr, None)
    for d in (a["formants"], a["status"], b["formants"], b["status"], changed, audio):
        d.free()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rank_processes_stitch_and_gather_over_rccl(tmp_path, world):
    """`world` rank processes (started as children, each on its own GPU): shard of ONE utterance -> vbx_analyze_frames_f64 ->
    vbx_comm_stitch_tracks_f64 -> vbx_gather_records_f64; rank 0 compares the gathered array with its own single-rank
    run of the whole recording, bit for bit.  At 8 ranks the tracker's state travels a chain of 7 hand-offs
    (ncclRecv -> stitch -> ncclSend per rank) and rank 0 receives 7 record blocks in one group.  Each size SKIPS on a box
    with fewer GPUs (this pool's boxes have one): the first multi-GPU lease validates the chain."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs: the multi-rank RCCL path (this pool's boxes have one)")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_rank_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    rep = json.load(open(os.path.join(str(tmp_path), "rank0.json")))
    assert rep["world"] == world and rep["bit_identical"] and rep["rows"] == rep["frames"]
    assert rep["changed"][0] == 0 and all(0 <= c <= 16 for c in rep["changed"][1:]), rep["changed"]
