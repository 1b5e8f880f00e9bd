import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg, o = g.load_package(), g.load_oracle()
vb = pkg.VoxBox(0)
N, H, SR = int(sys.argv[1]), int(sys.argv[2]), 48000.0
nf = 3000
ns = (nf - 1) * H + N
audio_d = vb.synth_speech(ns, sample_offset=5 * 48000)
han = vb.window(pkg.WINDOW_HANNING, N)
cand, cnt, st = vb.pitch(audio_d, SR, 0.2, 75.0, 600.0, kmax=4, frame_len=N, stride=H, n_frames=nf, window=han)
os.environ["VBX_PITCH_MFMA"] = "1"
vb2 = pkg.VoxBox(0)
a2 = vb2.synth_speech(ns, sample_offset=5 * 48000)
cand2, cnt2, st2 = vb2.pitch(a2, SR, 0.2, 75.0, 600.0, kmax=4, frame_len=N, stride=H, n_frames=nf, window=vb2.window(pkg.WINDOW_HANNING, N))
wh = o.window("hanning", N)
bad = 0
for t in range(0, nf):
    fr = audio_d.numpy_slice(t * H, N)
    es, ec, en = o.pitch(fr * wh, SR, 0.2, 75.0, 600.0)
    ok = abs(cand[t, 0, 0] - ec[0, 0]) <= 1e-4 * abs(ec[0, 0]) and abs(cand[t, 0, 1] - ec[0, 1]) <= 1e-4
    if not ok:
        bad += 1
        if bad <= 6:
            print("frame", t, "cnt", cnt[t], cnt2[t], en)
            print("  gpu fft   ", cand[t, :3].tolist())
            print("  gpu direct", cand2[t, :3].tolist())
            print("  oracle    ", ec[:3].tolist())
print("bad", bad, "of", nf)
