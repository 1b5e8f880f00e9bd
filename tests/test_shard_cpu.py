"""CPU tests of the multi-GPU host logic: frame-range split with the window halo,
utterance-aligned tracker segments, and the per-frame record gather to rank 0 -- exercised with
world_size 2 over gloo.  The per-shard compute stand-in here is the CPU oracle (allowed: tests
may call the oracle); on the GPU box the same helpers feed libvoxbox_hip (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, H, SR, P = 1200, 480, 48000.0, 12


def test_frame_range_partitions_everything(pkg):
    sh = pkg.shard
    for n_frames in (0, 1, 7, 8, 1000, 35999998):
        for world in (1, 2, 3, 8):
            ranges = [sh.frame_range(r, world, n_frames) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n_frames
            for (a, b), (c, d) in zip(ranges, ranges[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_sample_range_has_the_window_halo(pkg):
    sh = pkg.shard
    # rank g needs samples [start*H, (end-1)*H + N): a halo of N-H = 720 samples
    s0, s1 = sh.sample_range(100, 200, N, H)
    assert s0 == 100 * H and s1 == 199 * H + N
    nxt0, _ = sh.sample_range(200, 300, N, H)
    assert s1 - nxt0 == N - H
    assert pkg.frame_count(s1 - s0, N, H) == 100
    assert sh.sample_range(5, 5, N, H) == (5 * H, 5 * H)


def test_segment_aligned_ranges(pkg):
    sh = pkg.shard
    seg = np.arange(0, 10000, 1000)
    rr = sh.segment_aligned_ranges(4, seg, 10000)
    assert rr[0][0] == 0 and rr[-1][1] == 10000
    for (a, b), (c, d) in zip(rr, rr[1:]):
        assert b == c
    for a, b in rr:
        assert a % 1000 == 0 and (b % 1000 == 0 or b == 10000)
    ls = sh.local_segments(seg, 3000, 6000)
    assert list(ls) == [0, 1000, 2000]
    assert list(sh.local_segments(seg, 2500, 4200)) == [0, 500, 1500]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()
    o = g.load_oracle()
    synth = importlib.import_module(g.PKG_NAME + ".synth")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_samples = 2 * 48000 + 777
    F = pkg.frame_count(n_samples, N, H)
    seg = np.array([0, 60, 131], dtype=np.int64)
    ranges = pkg.shard.segment_aligned_ranges(world, seg, F)
    lo, hi = ranges[rank]
    s0, s1 = pkg.shard.sample_range(lo, hi, N, H)
    audio = synth.synth_speech(s1 - s0, sample_offset=s0)          # each rank generates ITS shard only
    w = o.window("hanning", N)
    lseg = pkg.shard.local_segments(seg, lo, hi)
    rec = np.zeros((hi - lo, 2 + 8 + 13), dtype=np.float64)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    est = None
    for t in range(hi - lo):
        fr = audio[t * H:t * H + N]
        _, c, _ = o.pitch(fr * w, SR, 0.2, 75.0, 600.0, cap=1)
        if t in lseg:
            est = est0.copy()
        _, est, _, _ = o.find_formants(fr, SR, P, est)
        rec[t, 0:2] = c[0]
        rec[t, 2:10] = est.reshape(-1)
        rec[t, 10:] = o.lpc(o.autocorrelate(fr * w, P + 1), P)
    counts = [b - a for a, b in ranges]
    full = pkg.shard.gather_records(torch.from_numpy(rec), counts, dst=0)
    if rank == 0:
        np.save(out_path, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_gather_equals_single_process(tmp_path, pkg, oracle):
    import importlib
    import torch.multiprocessing as mp
    import __graft_entry__ as g
    synth = importlib.import_module(g.PKG_NAME + ".synth")
    out = str(tmp_path / "gathered.npy")
    mp.start_processes(_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    # single-process reference over the whole recording
    n_samples = 2 * 48000 + 777
    audio = synth.synth_speech(n_samples)
    F = pkg.frame_count(n_samples, N, H)
    assert got.shape == (F, 23)
    w = oracle.window("hanning", N)
    seg = [0, 60, 131]
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    est = None
    for t in range(F):
        fr = audio[t * H:t * H + N]
        _, c, _ = oracle.pitch(fr * w, SR, 0.2, 75.0, 600.0, cap=1)
        if t in seg:
            est = est0.copy()
        _, est, _, _ = oracle.find_formants(fr, SR, P, est)
        exp = np.concatenate([c[0], est.reshape(-1), oracle.lpc(oracle.autocorrelate(fr * w, P + 1), P)])
        assert np.array_equal(got[t], exp), t


def test_library_shard_helpers_match_the_python_ones(pkg):
    """vbx_shard_range / vbx_shard_samples (what a C or Rust caller of the ABI uses) == shard.py."""
    sh = pkg.shard
    for n_frames in (0, 1, 7, 1000, 35999998):
        for world in (1, 2, 3, 8):
            assert [pkg.shard_range(n_frames, world, r) for r in range(world)] == \
                   [sh.frame_range(r, world, n_frames) for r in range(world)]
    for seg, n in ((np.arange(0, 10000, 1000), 10000), (np.array([0, 60, 131]), 197), (np.array([0]), 50), (None, 50),
                   (np.arange(0, 36000000, 1000), 36000000), (np.array([0]), 36000000), (np.array([0, 5, 2000, 2070, 4000]), 4100),
                   (None, 3), (np.array([0]), 5)):
        for world in (1, 2, 3, 4, 8):
            rr = sh.shard_ranges(world, seg, n)
            assert [pkg.shard_range(n, world, r, seg) for r in range(world)] == rr
            assert rr[0][0] == 0 and rr[-1][1] == n and all(b == c for (_, b), (c, _) in zip(rr, rr[1:]))
            for r in range(world):
                pl = pkg.shard_plan(n, world, r, seg)
                assert pl.as_dict() == sh.plan(n, world, r, seg)
                assert np.array_equal(pkg.shard_local_segments(pl, seg), sh.plan_local_segments(pl.as_dict(), seg))
                if r + 1 < world:                     # what one rank sends the next one expects
                    assert pl.continues_next == pkg.shard_plan(n, world, r + 1, seg).continues_prev
    # ONE utterance (what the reference's loop over a file is, tests/lib.rs:75-79) splits evenly; utterances that end on the
    # even cuts are not cut at all
    assert [pkg.shard_range(36_000_000, 8, r, np.array([0])) for r in range(8)] == [sh.frame_range(r, 8, 36_000_000) for r in range(8)]
    p1 = pkg.shard_plan(36_000_000, 8, 1, np.array([0])).as_dict()
    assert p1 == dict(lo=4_500_000, hi=9_000_000, warm=64, stop=4_500_064, continues_prev=1, continues_next=1)
    pa = pkg.shard_plan(36_000_000, 8, 3, np.arange(0, 36_000_000, 1000)).as_dict()
    assert pa["warm"] == 0 and pa["continues_prev"] == 0 and pa["continues_next"] == 0 and pa["lo"] % 1000 == 0
    assert pkg.shard_samples(100, 200, N, H) == sh.sample_range(100, 200, N, H)
    assert pkg.shard_samples(5, 5, N, H) == sh.sample_range(5, 5, N, H)


def _one_utterance_worker(rank, world, port, out_path, warm_frames, seg_list):
    """One rank of a recording whose utterances are cut by the rank boundaries: resonance rows of its frames [lo - warm, hi)
    (oracle), the track from the initial estimates, the state hand-off along the chain of ranks (gloo send / recv of the
    last formant row), the repair step (shard.stitch_rows = what tracker_stitch_kernel does), the gather."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()
    o = g.load_oracle()
    synth = importlib.import_module(g.PKG_NAME + ".synth")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = pkg.shard
    n_samples = 4 * 48000 + 777
    F = pkg.frame_count(n_samples, N, H)
    seg = None if seg_list is None else np.array(seg_list, dtype=np.int64)
    pl = sh.plan(F, world, rank, seg, warm_frames=warm_frames)
    if warm_frames == sh.WARM_FRAMES:
        assert pl == pkg.shard_plan(F, world, rank, seg).as_dict()
    first = pl["lo"] - pl["warm"]
    s0, s1 = sh.sample_range(first, pl["hi"], N, H)
    audio = synth.synth_speech(s1 - s0, sample_offset=s0 + 3 * 48000)        # each rank generates ITS shard only
    n = pl["hi"] - first
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    sk = o.soak(audio, N, H, 0, n, P, SR, o.SOAK_FORMANTS, n_threads=2)
    lseg = sh.plan_local_segments(pl, seg)
    rows = o.soak_track(sk["res"], sk["ff_status"], est0, lseg)               # tracked from a GUESS at frame lo - warm
    changed = 0
    if pl["continues_prev"]:
        state = torch.zeros(8, dtype=torch.float64)
        dist.recv(state, src=rank - 1)
        step = lambda st, t: (o.estimate_formants(st, sk["res"][t]) if sk["ff_status"][t] == 0 else st)
        changed = sh.stitch_rows(rows, pl["warm"], pl["stop"], state.numpy().reshape(4, 2), step)
    if pl["continues_next"]:
        dist.send(torch.from_numpy(rows[-1].reshape(-1).copy()), dst=rank + 1)
    rec = rows[pl["warm"]:].reshape(pl["hi"] - pl["lo"], 8)
    counts = [b - a for a, b in sh.shard_ranges(world, seg, F)]
    full = sh.gather_records(torch.from_numpy(np.ascontiguousarray(rec)), counts, dst=0)
    ch = [None] * world
    dist.all_gather_object(ch, changed)
    if rank == 0:
        np.save(out_path, full.numpy())
        np.save(out_path + ".changed.npy", np.array(ch))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,warm_frames,seg_list", [(2, 64, None), (3, 64, None), (3, 1, None), (2, 0, None),
                                                        (3, 64, [0, 150, 260]), (3, 3, [0, 131, 140, 300])])
def test_cut_utterances_track_like_a_single_process(tmp_path, pkg, oracle, world, warm_frames, seg_list):
    """SURVEY 8e "Exception": the tracker across rank boundaries.  One utterance (or utterances cut by the even split) over
    2 and 3 gloo ranks gives formant tracks bit-identical to the single-process sequential scan -- with the default warm-up
    (the guess is right, nothing is rewritten) and with a warm-up too short to be right (the repair step runs)."""
    import importlib
    import torch.multiprocessing as mp
    import __graft_entry__ as g
    synth = importlib.import_module(g.PKG_NAME + ".synth")
    out = str(tmp_path / "tracks.npy")
    mp.start_processes(_one_utterance_worker, args=(world, _free_port(), out, warm_frames, seg_list), nprocs=world, join=True,
                       start_method="spawn")
    got = np.load(out)
    changed = np.load(out + ".changed.npy")
    n_samples = 4 * 48000 + 777
    audio = synth.synth_speech(n_samples, sample_offset=3 * 48000)
    F = pkg.frame_count(n_samples, N, H)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    sk = oracle.soak(audio, N, H, 0, F, P, SR, oracle.SOAK_FORMANTS)
    exp = oracle.soak_track(sk["res"], sk["ff_status"], est0, None if seg_list is None else np.array(seg_list, dtype=np.int64))
    assert got.shape == (F, 8)
    assert np.array_equal(got, exp.reshape(F, 8))
    assert changed[0] == 0
    if warm_frames >= 64:
        assert np.all(changed == 0), changed          # the warmed-up guess was right: nothing rewritten
    if warm_frames <= 1 and seg_list is None:
        assert np.any(changed > 0), changed           # the repair path did run


def test_gather_plan_for_uneven_shards(pkg):
    """vbx_gather_plan = the transfer list vbx_gather_records_f64 posts (one grouped ncclRecv per sending peer on dst, one
    ncclSend on every peer with rows), checked on the host for world 2 / 3 / 8 with uneven and empty shards: the offsets
    tile the gathered array in rank order, every byte is received exactly once, and sends pair with receives."""
    REC = 36
    for rows in ([5, 3], [0, 4], [4, 0], [7, 0, 2], [1000, 1000, 999], [3, 1, 4, 1, 5, 9, 2, 6], [0, 0, 0, 0, 0, 0, 0, 8]):
        world = len(rows)
        for dst in (0, world - 1):
            plans = [pkg.gather_plan(rows, r, dst, REC) for r in range(world)]
            off_d, cnt_d, op_d = plans[dst]
            assert off_d[0] == 0 and np.array_equal(off_d[1:], np.cumsum(np.array(rows) * REC)[:-1])
            assert np.array_equal(cnt_d, np.array(rows) * REC)
            covered = np.zeros(sum(rows) * REC, dtype=np.int32)
            for r in range(world):
                off, cnt, op = plans[r]
                assert np.array_equal(off, off_d) and np.array_equal(cnt, cnt_d)      # every rank agrees on the layout
                if r == dst:
                    for q in range(world):
                        exp = pkg.GATHER_NONE if rows[q] == 0 else (pkg.GATHER_COPY if q == dst else pkg.GATHER_RECV)
                        assert op[q] == exp, (rows, dst, q)
                        if op[q] != pkg.GATHER_NONE:
                            covered[off[q]:off[q] + cnt[q]] += 1
                else:
                    sends = [q for q in range(world) if op[q] == pkg.GATHER_SEND]
                    assert sends == ([dst] if rows[r] > 0 else []), (rows, dst, r)
                    assert all(op[q] == pkg.GATHER_NONE for q in range(world) if q != dst)
                    # the matching receive exists on dst with the same element count
                    assert (op_d[r] == pkg.GATHER_RECV) == (rows[r] > 0) and cnt_d[r] == rows[r] * REC
            assert np.all(covered == 1)
    # the bench's shards: utterance-aligned ranges of an uneven recording
    seg = np.arange(0, 10_500, 1000)
    for world in (2, 3, 8):
        rr = [pkg.shard_range(10_500, world, r, seg) for r in range(world)]
        rows = [b - a for a, b in rr]
        off, cnt, _ = pkg.gather_plan(rows, 0, 0, REC)
        assert [int(o) // REC for o in off] == [a for a, _ in rr] and int(off[-1] + cnt[-1]) == 10_500 * REC
    with pytest.raises(pkg.VoxBoxError):
        pkg.gather_plan([1, -1], 0, 0, REC)
    assert pkg.comm_live_count() == 0


def test_bench_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` with no RANK in the environment starts 2 rank processes before touching the GPU,
    relays rank 0's JSON line and returns their status (dry run: the ranks only rendezvous over gloo)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["VBX_BENCH_DRY_RUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["dry_run"] and line["n_gpus"] == 2 and line["gpus_arg"] == 2
    assert sorted(x[0] for x in line["ranks"]) == [0, 1] and sorted(x[1] for x in line["ranks"]) == [0, 1]
    assert len({x[2] for x in line["ranks"]}) == 1
    # control plane over gloo, ONE RCCL communicator per rank (the library's), no torch nccl process group, and the
    # gather's transfer list: rank 0 keeps its rows in place and receives rank 1's, rank 1 sends to rank 0
    assert line["control_plane"] == "gloo" and line["rccl_comms_per_rank_planned"] == 1 and line["torch_nccl_process_groups"] == 0
    gp = line["gather_plan"]
    assert gp["ops_by_rank"] == [[3, 1], [2, 0]] and gp["offsets"] == [0, gp["rows"][0] * gp["record_doubles"]]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "init_process_group(\"nccl\"" not in src and "batch_isend_irecv" not in src      # no second RCCL instance, no fallback transport
    # a failing rank makes the launcher fail (no GPU here: the ranks refuse to run)
    env.pop("VBX_BENCH_DRY_RUN")
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0


# ------------------------------------------------------------------------------------------------------------------------
# Round 6: bench.py's OWN N > 1 step (pipeline_buffers + pipeline_step: wait -> analyze -> stitch -> gather, double-buffered slots,
# the warm-up rows' offset into the send buffer) at world 8 over gloo, with test doubles for the two things a CPU box lacks: the
# context (its analyze_frames is the oracle's walk of the rank's frames, tracked from a GUESS at the first warm-up frame, as the
# library does) and the communicator (gloo send / recv of the same rows the library moves with ncclSend / ncclRecv; the repair step
# is shard.stitch_rows = what tracker_stitch_kernel does).  The doubles live HERE; the product never sees them.
# ------------------------------------------------------------------------------------------------------------------------
REC_DOUBLES = 36      # pitch (2) | formants (8) | lpc (13) | mfcc (13): vbx_record_doubles of the bench's parameters


def _view(ptr, rows, ld):
    """[rows, ld] doubles at a raw address (what the C ABI is handed: rec[b].data_ptr() + an offset)"""
    import ctypes
    return np.ctypeslib.as_array((ctypes.c_double * (rows * ld)).from_address(int(ptr))).reshape(rows, ld)


class _OracleContext:
    """Stands in for VoxBox in bench.pipeline_step: analyze_frames fills the record rows of frames [first, first + n_frames)."""

    def __init__(self, o, sh, pkg, first_frame, rank):
        self.o, self.sh, self.pkg, self.first, self.rank, self.calls = o, sh, pkg, first_frame, rank, 0
        self._cache = None

    def analyze_frames(self, audio, params, seg_start=None, frame_len=N, stride=H, n_frames=0, out=None, record_ld=0, status=None):
        self.calls += 1
        if self._cache is None:                     # the analysis does not depend on the step: computed once, written every step
            est0 = np.array([[f, 1.0] for f in self.pkg.MALE_FORMANT_ESTIMATES])
            sk = self.o.soak(audio.numpy(), frame_len, stride, 0, n_frames, P, SR, self.o.SOAK_FORMANTS, n_threads=1)
            self._cache = (sk, self.o.soak_track(sk["res"], sk["ff_status"], est0, seg_start))
        sk, rows = self._cache
        rec = out.numpy()
        rec[:] = -1.0                               # stale rows of the slot's previous use must not survive
        rec[:, 0] = self.first + np.arange(n_frames)          # the frame's GLOBAL index where the pitch frequency would be
        rec[:, 1] = self.rank
        rec[:, 2:10] = rows.reshape(n_frames, 8)
        status.zero_()


class _GlooComm:
    """Stands in for voxbox.Comm: same calls, same raw pointers, torch.distributed (gloo) underneath."""

    def __init__(self, dist, torch, world, rank, sh, o, ctx):
        self.dist, self.torch, self.world, self.rank, self.sh, self.o, self.ctx = dist, torch, world, rank, sh, o, ctx
        self.log, self.changed = [], 0

    def wait(self, slot):
        self.log.append(("wait", slot))

    def stitch_tracks(self, formants_ptr, n_frames, ld, plan, changed=None, slot=0):
        self.log.append(("stitch", slot))
        pl = plan.as_dict()
        rows = _view(formants_ptr - 16, n_frames, ld)         # the record rows; formants are doubles 2 .. 9
        sk, _ = self.ctx._cache
        if pl["continues_prev"]:
            state = self.torch.zeros(8, dtype=self.torch.float64)
            self.dist.recv(state, src=self.rank - 1)
            trk = rows[:, 2:10].reshape(n_frames, 4, 2).copy()
            step = lambda st, t: (self.o.estimate_formants(st, sk["res"][t]) if sk["ff_status"][t] == 0 else st)
            self.changed += self.sh.stitch_rows(trk, pl["warm"], pl["stop"], state.numpy().reshape(4, 2), step)
            rows[:, 2:10] = trk.reshape(n_frames, 8)
        if pl["continues_next"]:
            self.dist.send(self.torch.from_numpy(rows[-1, 2:10].copy()), dst=self.rank + 1)

    def gather_records(self, local_ptr, counts, row_doubles, dst=0, out=None, slot=0):
        self.log.append(("gather", slot))
        mine = self.torch.from_numpy(_view(local_ptr, counts[self.rank], row_doubles).copy())
        if self.rank == dst:
            off = 0
            for r in range(self.world):
                if r != dst:
                    buf = self.torch.empty((counts[r], row_doubles), dtype=self.torch.float64)
                    self.dist.recv(buf, src=r)
                    out[off:off + counts[r]] = buf
                else:
                    assert out.data_ptr() + off * row_doubles * 8 == local_ptr      # rank 0's rows are written in place
                off += counts[r]
        else:
            self.dist.send(mine, dst=dst)


def _bench_step_worker(rank, world, port, out_path, F, steps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import importlib.util
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()
    o = g.load_oracle()
    synth = importlib.import_module(g.PKG_NAME + ".synth")
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # run_rank's shard geometry, verbatim: the recording is ONE utterance of world * F frames
    plan = pkg.shard_plan(world * F, world, rank, None)
    lo, hi, warm = plan.lo, plan.hi, plan.warm
    assert hi - lo == F
    seg = pkg.shard_local_segments(plan, None)
    s0, s1 = pkg.shard_samples(lo - warm, hi, N, H)
    audio = torch.from_numpy(synth.synth_speech(s1 - s0, sample_offset=s0))
    FA = F + warm
    counts = [F] * world
    ctx = _OracleContext(o, pkg.shard, pkg, lo - warm, rank)
    comm = _GlooComm(dist, torch, world, rank, pkg.shard, o, ctx)
    rec, gathered = bench.pipeline_buffers(torch, "cpu", world, rank, F, FA, REC_DOUBLES)
    assert len(rec) == 2 and (rank != 0 or rec[0].data_ptr() == gathered[0].data_ptr())
    st3 = torch.empty((3, FA), dtype=torch.int32)
    stitch = bool(plan.continues_prev or plan.continues_next)
    step = bench.pipeline_step(ctx, comm, audio, None, seg, N, H, FA, warm, REC_DOUBLES, rec, gathered, st3, counts, plan, stitch)
    for i in range(steps):
        step(i)
    per_step = [("wait", None), ("stitch", None), ("gather", None)] if stitch else [("wait", None), ("gather", None)]
    assert [k for k, _ in comm.log] == [k for k, _ in per_step] * steps, comm.log
    assert [s for _, s in comm.log] == [i % 2 for i in range(steps) for _ in per_step], comm.log       # the two slots alternate
    assert ctx.calls == steps
    ch = [None] * world
    dist.all_gather_object(ch, comm.changed)
    if rank == 0:
        np.save(out_path, np.stack([g_.numpy() for g_ in gathered]))
        np.save(out_path + ".changed.npy", np.array(ch))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,F,steps", [(8, 80, 3), (2, 150, 2)])
def test_bench_pipeline_step_at_world_8_over_gloo(tmp_path, pkg, oracle, world, F, steps):
    """bench.py's own step() ordering and buffer arithmetic at world 8: after `steps` steps BOTH slots of rank 0's
    gathered array hold, for every frame of the 8 x F-frame utterance in order, its global index, its rank, and the formant track of
    the single-process sequential scan -- bit for bit.  Wrong warm-row offsets, a slot mix-up or a gather that lands rows in the wrong
    place all show here."""
    import importlib
    import torch.multiprocessing as mp
    import __graft_entry__ as g
    synth = importlib.import_module(g.PKG_NAME + ".synth")
    out = str(tmp_path / "gathered.npy")
    mp.start_processes(_bench_step_worker, args=(world, _free_port(), out, F, steps), nprocs=world, join=True, start_method="spawn")
    got = np.load(out)                                  # [2 slots, world * F, REC]
    total = world * F
    audio = synth.synth_speech((total - 1) * H + N, sample_offset=0)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    sk = oracle.soak(audio, N, H, 0, total, P, SR, oracle.SOAK_FORMANTS)
    exp = oracle.soak_track(sk["res"], sk["ff_status"], est0, None).reshape(total, 8)
    slots_used = {i % 2 for i in range(steps)}
    for b in sorted(slots_used):
        assert np.array_equal(got[b][:, 0], np.arange(total)), f"slot {b}: rows out of place"
        assert np.array_equal(got[b][:, 1], np.repeat(np.arange(world), F)), f"slot {b}: rows from the wrong rank"
        assert np.array_equal(got[b][:, 2:10], exp), f"slot {b}: tracks differ from the single-process scan"
    assert np.load(out + ".changed.npy")[0] == 0


def test_the_launcher_kills_ranks_that_ignore_terminate():
    """bench.supervise: when one rank dies the launcher terminate()s the others and, after a grace period,
    kill()s whatever ignored that -- a rank stuck inside a collective does not run signal handlers -- and returns non-zero."""
    import importlib.util
    import subprocess
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stubborn = "import signal, time, sys\nsignal.signal(signal.SIGTERM, signal.SIG_IGN)\nprint('{\"rank0\": 1}', flush=True)\ntime.sleep(120)\n"
    dies = "import sys, time\ntime.sleep(0.5)\nsys.exit(5)\n"
    procs = [subprocess.Popen([sys.executable, "-c", stubborn], stdout=subprocess.PIPE, text=True),
             subprocess.Popen([sys.executable, "-c", dies])]
    t0 = time.monotonic()
    rc = bench.supervise(procs, grace_s=2.0, relay=None)
    dt = time.monotonic() - t0
    assert rc == 1 and dt < 30.0, (rc, dt)
    assert procs[0].returncode == -9 and procs[1].returncode == 5, [p.returncode for p in procs]
    # all ranks fine: 0, and rank 0's line is relayed
    import io
    ok = [subprocess.Popen([sys.executable, "-c", "print('{\"v\": 1}')"], stdout=subprocess.PIPE, text=True),
          subprocess.Popen([sys.executable, "-c", "pass"])]
    buf = io.StringIO()
    assert bench.supervise(ok, relay=buf) == 0 and buf.getvalue().strip() == '{"v": 1}'


def test_bench_dry_run_at_world_8_plans_the_chain_of_seven_hand_offs():
    """`python bench.py --gpus 8` (dry run: rendezvous over gloo, host arithmetic only): eight ranks, BASELINE config 5's 100 h as ONE
    utterance split evenly -- every rank but the first warms its tracker up over 64 frames and continues its predecessor's track,
    every rank but the last passes its own on; the gather's transfer list has rank 0 receive seven shards in rank order."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["VBX_BENCH_DRY_RUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["dry_run"] and line["n_gpus"] == 8 and sorted(x[0] for x in line["ranks"]) == list(range(8))
    F = 4_500_000
    plans = line["shard_plans"]
    assert [p["lo"] for p in plans] == [r_ * F for r_ in range(8)] and [p["hi"] for p in plans] == [(r_ + 1) * F for r_ in range(8)]
    assert [p["warm"] for p in plans] == [0] + [64] * 7
    assert [p["continues_prev"] for p in plans] == [0] + [1] * 7 and [p["continues_next"] for p in plans] == [1] * 7 + [0]
    gp = line["gather_plan"]
    assert gp["rows"] == [F] * 8 and gp["offsets"] == [r_ * F * gp["record_doubles"] for r_ in range(8)]
    assert gp["ops_by_rank"][0] == [3] + [1] * 7                              # rank 0: its own rows in place, a receive per peer
    for r_ in range(1, 8):
        assert gp["ops_by_rank"][r_] == [2] + [0] * 7                          # every peer: one send to rank 0
