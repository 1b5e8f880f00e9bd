"""Frame-range sharding of one long recording across the GPUs of a node.

Frames are independent on the whole path except for the formant tracker, whose state is
reset at utterance boundaries, so the batch splits by contiguous frame ranges with no
data-path collective; the only exchange is one gather of fixed-size per-frame records to
rank 0 (RCCL over xGMI when the tensors live on the GPUs, gloo in the CPU tests).
"""
import numpy as np


def frame_range(rank, world, n_frames):
    """Contiguous split [lo, hi) of n_frames over `world` ranks (first ranks take the remainder)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def segment_aligned_ranges(world, seg_start, n_frames):
    """Split at utterance boundaries so every tracker segment lives on exactly one rank.
    seg_start: ascending frame indices (seg_start[0] == 0).  Returns [(lo, hi)] per rank."""
    seg = np.asarray(seg_start, dtype=np.int64)
    bounds = np.append(seg, n_frames)
    out = []
    prev = 0
    for r in range(world):
        target = frame_range(r, world, n_frames)[1]
        # first boundary >= target (the last rank always ends at n_frames)
        idx = int(np.searchsorted(bounds, target, side="left"))
        hi = int(bounds[min(idx, len(bounds) - 1)]) if r < world - 1 else n_frames
        hi = max(hi, prev)
        out.append((prev, hi))
        prev = hi
    return out


WARM_FRAMES = 64          # == VBX_SHARD_WARM_FRAMES


def shard_ranges(world, seg_start, n_frames):
    """vbx_shard_range for every rank: the even split; a cut moves up to an utterance start that lies within 1/32 of a
    shard after it, otherwise it stays where it is -- inside the utterance (the track is carried across, `plan`)."""
    bounds = None if seg_start is None else np.asarray(seg_start, dtype=np.int64)
    slack = n_frames // world // 32
    cuts = [0]
    for r in range(world):
        if r == world - 1:
            c = n_frames
        else:
            c = frame_range(r, world, n_frames)[1]
            if bounds is not None and bounds.size:
                idx = int(np.searchsorted(bounds, c, side="left"))
                if idx < bounds.size and bounds[idx] <= c + slack and bounds[idx] <= n_frames:
                    c = int(bounds[idx])
        cuts.append(max(c, cuts[-1]))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def plan(n_frames, world, rank, seg_start=None, warm_frames=WARM_FRAMES):
    """vbx_shard_plan as a dict: lo, hi, warm (extra leading frames [lo - warm, lo) that warm the tracker up), stop (index
    from frame lo - warm at which the utterance holding frame lo ends inside the shard), continues_prev / continues_next."""
    seg = np.array([0], dtype=np.int64) if seg_start is None else np.asarray(seg_start, dtype=np.int64)
    lo, hi = shard_ranges(world, seg_start, n_frames)[rank]
    start_of = lambda f: int(seg[np.searchsorted(seg, f, side="right") - 1]) if seg.size else 0
    continued = lambda c: 0 < c < n_frames and c - start_of(c) > warm_frames
    out = dict(lo=lo, hi=hi, warm=0, stop=0, continues_prev=0, continues_next=0)
    if hi <= lo:
        return out
    out["warm"] = min(lo - start_of(lo), warm_frames)
    out["continues_prev"] = int(continued(lo))
    out["continues_next"] = int(continued(hi))
    after = seg[seg > lo]
    nxt = min(int(after[0]), n_frames) if after.size else n_frames
    out["stop"] = min(nxt, hi) - (lo - out["warm"])
    return out


def plan_local_segments(pl, seg_start=None):
    """vbx_shard_local_segments: utterance starts of frames [lo - warm, hi), re-based (first entry 0)."""
    first = pl["lo"] - pl["warm"]
    if seg_start is None:
        return np.array([0], dtype=np.int64)
    seg = np.asarray(seg_start, dtype=np.int64)
    return np.concatenate([[0], seg[(seg > first) & (seg < pl["hi"])] - first]).astype(np.int64)


def stitch_rows(rows, first, stop, state_in, step):
    """The repair step of vbx_track_stitch_f64 on host arrays (the CPU tests' stand-in for the kernel; `step(state, t)` is
    one tracker step on local frame t and returns the new state).  rows: [n, n_est, 2] formant rows of frames
    [lo - warm, hi) tracked from a guess; state_in: the row the previous rank ends with.  Rows [first, stop) are rewritten
    until the redone scan meets a row it already holds.  Returns the number of rows rewritten."""
    state = np.array(state_in, dtype=np.float64, copy=True)
    if first > 0 and rows[first - 1].tobytes() == state.tobytes():
        return 0
    n = 0
    for t in range(first, stop):
        state = step(state, t)
        if rows[t].tobytes() == state.tobytes():
            break
        rows[t] = state
        n += 1
    return n


def sample_range(lo, hi, frame_len, hop):
    """Samples [s0, s1) a rank needs for frames [lo, hi): includes the (frame_len - hop) halo."""
    if hi <= lo:
        return lo * hop, lo * hop
    return lo * hop, (hi - 1) * hop + frame_len


def local_segments(seg_start, lo, hi):
    """Tracker segment starts of the frames [lo, hi), re-based to the shard (first entry 0)."""
    seg = np.asarray(seg_start, dtype=np.int64)
    inside = seg[(seg > lo) & (seg < hi)] - lo
    return np.concatenate([[0], inside]).astype(np.int64)


def gather_records(local, counts, dst=0, group=None):
    """Gathers per-frame record tensors [n_local, rec] from every rank to `dst` in rank order.
    `counts[r]` = frames owned by rank r.  Uses torch.distributed (nccl == RCCL on ROCm, or
    gloo).  Returns the concatenated [sum(counts), rec] tensor on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rec = local.shape[1:]
    if world == 1:
        return local
    if dist.get_backend(group) == "nccl":
        # one grouped send/recv set: every peer's payload crosses its own xGMI link to dst
        if rank == dst:
            parts = [local if r == dst else torch.empty((counts[r],) + tuple(rec), dtype=local.dtype, device=local.device)
                     for r in range(world)]
            ops = [dist.P2POp(dist.irecv, parts[r], r, group) for r in range(world) if r != dst and counts[r] > 0]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return torch.cat(parts, dim=0)
        if counts[rank] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), dst, group)]):
                w.wait()
        return None
    # gloo (CPU tests): pad to the largest shard and use gather
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(rec), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, bufs, dst=dst, group=group)
        return torch.cat([bufs[r][: counts[r]] for r in range(world)], dim=0)
    dist.gather(pad, None, dst=dst, group=group)
    return None
