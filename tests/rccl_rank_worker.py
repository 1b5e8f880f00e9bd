"""One rank process of tests/test_gpu_shard.py::test_rank_processes_stitch_and_gather_over_rccl (started as a child,
RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment).  Control plane: gloo; data plane: the library's RCCL
communicator (the state hand-off of the formant tracker and the record gather)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, H, SR, P = 1200, 480, 48000.0, 12
F = 20_000
OFFSET = 5 * 48000


def main(outdir):
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = g.load_package()
    torch.cuda.set_device(local)
    vb = pkg.VoxBox(local)
    ids = [pkg.comm_unique_id()] if rank == 0 else [None]
    dist.broadcast_object_list(ids, src=0)
    comm = pkg.Comm(vb, ids[0], world, rank)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    params = pkg.AnalysisParams.make(SR, pitch=(0.2, 75.0, 600.0), lpc_order=P, formant_order=P, est_init=est0, mfcc=(13, 100.0, 8000.0))
    REC = int(vb.L.vbx_record_doubles(params))
    pl = pkg.shard_plan(F, world, rank, None)              # ONE utterance: the even split cuts it
    first, n = pl.lo - pl.warm, pl.hi - pl.lo + pl.warm
    audio = vb.synth_speech((n - 1) * H + N, sample_offset=OFFSET + first * H)
    rec, st3, changed = vb.empty((n, REC)), vb.empty((3, n), np.int32), vb.empty(1, np.int32)
    vb.analyze_frames(audio, params, frame_len=N, stride=H, n_frames=n, out=rec, record_ld=REC, status=st3)
    comm.stitch_tracks(rec.ptr + 2 * 8, n, REC, pl, changed, slot=0)
    counts = [pkg.shard_plan(F, world, r, None).hi - pkg.shard_plan(F, world, r, None).lo for r in range(world)]
    gathered = vb.empty((F, REC)) if rank == 0 else None
    comm.gather_records(rec.ptr + pl.warm * REC * 8, counts, REC, 0, out=gathered, slot=0)
    comm.sync()
    vb.sync()
    ch = [None] * world
    dist.all_gather_object(ch, int(changed.numpy()[0]))
    if rank == 0:
        whole_audio = vb.synth_speech((F - 1) * H + N, sample_offset=OFFSET)
        whole, _ = vb.analyze_frames(whole_audio, params, frame_len=N, stride=H, n_frames=F)
        got = gathered.numpy()
        with open(os.path.join(outdir, "rank0.json"), "w") as f:
            json.dump({"bit_identical": bool(np.array_equal(got, whole)), "rows": int(got.shape[0]), "frames": F, "changed": ch,
                       "world": world}, f)
    dist.barrier()
    comm.close()
    vb.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
