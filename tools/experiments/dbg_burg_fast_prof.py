"""Per-kernel times of the one-pass Burg under rocprofv3 --kernel-trace --stats: python tools/experiments/dbg_burg_fast_prof.py N H F [pcm]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
vb = pkg.VoxBox()
N, H, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
audio = vb.synth_speech((F - 1) * H + N, sample_offset=3 * 48000)
w = vb.window(2, N)
out = (vb.empty((F, 12)), vb.empty(F, np.int32))
for rep in range(5):
    vb.lpc_praat(audio, 12, frame_len=N, stride=H, n_frames=F, window=w, out=out)
vb.sync()
print("direct count", vb.last_burg_direct_count())
