import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
N, H, F = 1200, 480, 300
a = vb.synth_speech((F - 1) * H + N, sample_offset=9 * 48000); x = a.numpy()
pcm = np.clip(np.rint(x / np.max(np.abs(x)) * 0.9 * 32767.0), -32768, 32767).astype(np.int16)
wide = vb.pcm16_to_f64(pcm)
params = pkg.AnalysisParams.make(48000.0, lpc_order=10)
print(params.columns())
r1, s1 = vb.analyze_frames(wide, params, frame_len=N, stride=H, n_frames=F)
r2, s2 = vb.analyze_frames(wide, params, frame_len=N, stride=H, n_frames=F)
r3, s3 = vb.analyze_frames_pcm16(pcm, params, frame_len=N, stride=H)
r4, s4 = vb.analyze_frames_pcm16(pcm, params, frame_len=N, stride=H)
for name, (p, q) in {"f64 vs f64": (r1, r2), "pcm vs f64": (r3, r1), "pcm vs pcm": (r3, r4)}.items():
    d = p != q
    print(name, "rows", int(d.any(axis=1).sum()), "cols", np.nonzero(d.any(axis=0))[0])
    if d.any():
        t = np.nonzero(d.any(axis=1))[0][0]; c = np.nonzero(d[t])[0]
        print("  row", t, "cols", c, p[t, c][:4], q[t, c][:4])
