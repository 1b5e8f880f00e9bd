#!/usr/bin/env python3
"""The chunked tracker scan against the sequential one on millions of random frames: rows that forget the state quickly,
rows that hardly ever do, silences, skipped frames, utterances from 1 frame to the whole batch.  Every output row must be
bit-identical.  usage (GPU box): python3 tools/soak_tracker.py [frames_per_case=1000000] [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def rows(rng, F, n_res, lo, hi, zero_rows):
    k = rng.integers(lo, hi + 1, F)
    k[rng.random(F) < zero_rows] = 0
    f = np.sort(rng.uniform(60.0, 8000.0, (F, n_res)), axis=1)
    b = rng.uniform(20.0, 900.0, (F, n_res))
    mask = np.arange(n_res)[None, :] < k[:, None]
    return np.stack([np.where(mask, f, 0.0), np.where(mask, b, 0.0)], axis=2)


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "soak_tracker.json")
    pkg = g.load_package()
    vb = pkg.VoxBox(0)
    rng = np.random.default_rng(2026)
    est4 = np.array([[f, 80.0] for f in pkg.MALE_FORMANT_ESTIMATES])
    cases = []
    for name, lo, hi, zr, n_est, segs in (
            ("rich rows, one utterance", 4, 7, 0.0, 4, "one"),
            ("sparse rows (0-2 resonances, 30 % empty), one utterance", 0, 2, 0.3, 4, "one"),
            ("mixed rows, utterances of 1..5000 frames, 2 % skipped frames", 0, 6, 0.1, 4, "random"),
            ("mixed rows, 6 estimates, utterances of 1..300 frames and one of 200000", 0, 7, 0.05, 6, "short+long"),
            ("single-resonance rows, 1 estimate, one utterance", 1, 1, 0.2, 1, "one")):
        res = rows(rng, F, 8, lo, hi, zr)
        est0 = np.concatenate([est4, [[5000.0, 100.0], [6500.0, 120.0]]])[:n_est]
        status = None
        if segs == "one":
            seg = None
        elif segs == "random":
            cuts = np.unique(np.concatenate([[0], np.cumsum(rng.integers(1, 5000, F // 1000))]))
            seg = cuts[cuts < F].astype(np.int64)
            status = (rng.random(F) < 0.02).astype(np.int32) * 2
        else:
            a = np.cumsum(rng.integers(1, 300, F // 100))
            a = a[a < F - 200000]
            seg = np.unique(np.concatenate([[0], a])).astype(np.int64)       # the last utterance runs to the end: >= 200000 frames
        outs = {}
        for mode in ("0", "1"):
            os.environ["VBX_TRACKER_CHUNKED"] = mode
            outs[mode] = vb.estimate_formants(res, est0, seg_start=seg, frame_status=status)
        diff = np.any(outs["0"].view(np.uint64) != outs["1"].view(np.uint64), axis=(1, 2))
        cases.append({"case": name, "frames": F, "utterances": 1 if seg is None else int(seg.size), "rows_that_differ": int(diff.sum())})
        print(json.dumps(cases[-1]), flush=True)
    summary = {"cases": cases, "rows_that_differ": int(sum(c["rows_that_differ"] for c in cases))}
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(summary, open(out_path, "w"), indent=1)
    print(json.dumps({"total_rows_that_differ": summary["rows_that_differ"]}))
    vb.close()


if __name__ == "__main__":
    main()
