#!/usr/bin/env python3
"""The full pipeline (vbx_analyze_frames_f64, everything on, one utterance) at other frame shapes: bench.py's pipeline_shapes on
its own.  usage: python3 tools/experiments/pipeline_shapes.py [hours=1] [N:hop ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

graft = bench.graft
pkg = graft.load_package()
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[2:]] or bench.PIPELINE_SHAPES
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ts = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ts)
vb = pkg.VoxBox(0, ts.cuda_stream)
ns = int(hours * 3600 * bench.SR) + 4096
audio = torch.empty(ns, dtype=torch.float64, device=dev)
vb.synth_speech(ns, sample_offset=0, sample_rate=bench.SR, out=audio)
for r in bench.pipeline_shapes(vb, torch, dev, pkg, audio, hours=hours, shapes=shapes):
    print("%d:%d %.2f M/s %s" % (r["frame_len"], r["hop"], r["value"] / 1e6, json.dumps(r["kernels_ms"])), flush=True)
vb.close()
