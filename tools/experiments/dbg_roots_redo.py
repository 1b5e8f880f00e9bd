import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox()
F=300000; N,H=512,512
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
audio = vb.synth_speech((F - 1) * H + N, sample_offset=3 * 48000)
seg = np.arange(0, F, 1000, dtype=np.int64)
os.environ["VBX_ROOTS_DIRECT"]="0"
a = vb.find_formants(audio, 48000.0, 12, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F)
print("redo", vb.last_roots_direct_count())
os.environ["VBX_ROOTS_DIRECT"]="1"
b = vb.find_formants(audio, 48000.0, 12, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F)
same = np.all(a["res"]==b["res"],axis=(1,2)) & (b["count"]>0)
idx=np.nonzero(same)[0]
print("bitwise same rows", idx.size, idx[:40])
np.save("/root/repo/gpurun_out/redo_coeffs.npy", a["coeffs"][idx] if a.get("coeffs") is not None else np.zeros(0))
print(a.keys())
