"""find_formants on a million 512-sample frames in utterances of `seg` frames: the time-sliced scan (equal short utterances,
VBX_TRACKER_CHUNKED unset / 0) against the chunked scan (VBX_TRACKER_CHUNKED=1).  usage: python tools/experiments/ff_short_utterances.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox()
F, N = 1_000_000, 512
audio = vb.synth_speech(F * N, sample_offset=3 * 48000)
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
ff = {"formants": vb.empty((F, 4, 2)), "res": None, "count": None, "coeffs": None, "status": vb.empty(F, np.int32)}
for seg_len in (64, 128, 256, 383, 1000):
    seg = np.arange(0, F, seg_len, dtype=np.int64)
    row = []
    for mode in (None, "1"):
        if mode is None: os.environ.pop("VBX_TRACKER_CHUNKED", None)
        else: os.environ["VBX_TRACKER_CHUNKED"] = mode
        best = 1e9
        for _ in range(4):
            vb.timer_begin()
            vb.find_formants(audio, 48000.0, 12, est0, seg_start=seg, frame_len=N, stride=N, n_frames=F, out=ff)
            best = min(best, vb.timer_end())
        row.append(best)
    print(f"utterances of {seg_len}: default path {row[0]:.2f} ms, chunked scan {row[1]:.2f} ms", flush=True)
