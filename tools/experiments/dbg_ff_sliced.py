import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
vb = pkg.VoxBox(0)
seg_len, n_seg = 256, 260
N, H, P = 512, 160, 12
F = seg_len * n_seg - 100
seg = np.arange(0, seg_len * n_seg, seg_len, dtype=np.int64)
audio = vb.synth_speech((F - 1) * H + N, sample_offset=7 * 48000)
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
os.environ.pop("VBX_TRACKER_CHUNKED", None)
a = vb.find_formants(audio, 48000.0, P, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F)
os.environ["VBX_TRACKER_CHUNKED"] = "1"
b = vb.find_formants(audio, 48000.0, P, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F)
for k in ("status", "count", "coeffs", "res", "formants"):
    d = a[k] != b[k]
    rows = np.nonzero(d.reshape(F, -1).any(axis=1))[0]
    print(k, rows.size, rows[:10], rows[-5:] if rows.size else "")
    if rows.size and k in ("res","coeffs"):
        t = rows[0]; print(" a", a[k][t].ravel()[:8]); print(" b", b[k][t].ravel()[:8])
