#!/usr/bin/env python3
"""BASELINE config 4 at full size against the oracle: vbx_find_formants_f64 over F dense frames in utterances of `seg` frames
(Burg -> roots -> resonances -> the tracker carried from frame to frame, reset at every utterance start) against the CPU
oracle's frame loop over EVERY utterance (host threads over utterances).  Formant Hz within 1e-4 relative, statuses exact.
usage (GPU box): python3 tools/soak_formant_tracks.py [frames=1000000] [seg=1000] [out.json] [frame_len=512]"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

SR, P = 48000.0, 12


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    seg_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "soak_formant_tracks.json")
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    pkg, o = g.load_package(), g.load_oracle()
    vb = pkg.VoxBox(0)
    try:
        workers = len(os.sched_getaffinity(0))
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            workers = max(1, min(workers, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError, AttributeError):
        workers = os.cpu_count() or 1
    audio_d = vb.synth_speech(F * N)
    est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    seg = np.arange(0, F, seg_len, dtype=np.int64)
    got = vb.find_formants(audio_d, SR, P, est0, seg_start=seg, frame_len=N, stride=N, n_frames=F, want=("formants", "status"))
    tracks, status = got["formants"], got["status"]

    def one(u):
        lo, hi = int(seg[u]), int(seg[u + 1]) if u + 1 < seg.size else F
        x = audio_d.numpy_slice(lo * N, (hi - lo) * N)
        est = est0.copy()
        bad_status = bad_hz = 0
        for t in range(lo, hi):
            s, est, _, _ = o.find_formants(x[(t - lo) * N:(t - lo + 1) * N], SR, P, est)
            bad_status += int(s != status[t])
            bad_hz += int(not np.all(np.abs(tracks[t, :, 0] - est[:, 0]) <= 1e-4 * np.abs(est[:, 0])))
        return bad_status, bad_hz

    with ThreadPoolExecutor(workers) as ex:
        res = list(ex.map(one, range(seg.size)))
    summary = {"frames": F, "frame_len": N, "utterances": int(seg.size), "frames_per_utterance": seg_len, "oracle_threads": workers,
               "frames_whose_status_differs": int(sum(r[0] for r in res)),
               "frames_with_a_formant_beyond_1e-4": int(sum(r[1] for r in res)),
               "frames_with_nonzero_status": int(np.count_nonzero(status))}
    print(json.dumps(summary))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(summary, open(out_path, "w"), indent=1)
    vb.close()


if __name__ == "__main__":
    main()
