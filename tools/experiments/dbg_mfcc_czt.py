import os, sys, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package(); o = g.load_oracle()
vb = pkg.VoxBox(0)
SR = 48000.0
audio_d = vb.synth_speech(3 * 48000, sample_offset=2 * 48000); audio = audio_d.numpy()
worst = {}
for n, k, lo, hi, sr in ((1103, 13, 100., 8000., 44100.), (1102, 13, 100., 8000., 44100.), (1600, 13, 100., 8000., 48000.), (3000, 13, 100., 8000., 48000.),
                         (601, 13, 100., 8000., 48000.), (997, 20, 50., 6000., 22050.), (2049, 26, 133., 6855., 22050.), (3301, 13, 100., 8000., 48000.),
                         (700, 40, 0., 10000., 48000.), (256, 13, 100., 4000., 16000.), (1200, 13, 100., 8000., 48000.), (2047, 13, 100., 3000., 48000.)):
    hop = n // 3
    F = min((audio.size - n) // hop + 1, 40)
    w = o.window("hanning", n)
    han = vb.window(pkg.WINDOW_HANNING, n)
    m, st = vb.mfcc(audio_d, k, (lo, hi), sr, frame_len=n, stride=hop, n_frames=F, window=han)
    err = 0.0
    for t in range(F):
        s, e = o.mfcc(audio[t*hop:t*hop+n] * w, k, lo, hi, sr)
        assert s == st[t], (n, t, s, st[t])
        if s == 0:
            err = max(err, float(np.max(np.abs(m[t] - e) / np.maximum(np.abs(e), 1e-6 * np.max(np.abs(e))))))
    worst[n] = err
print("CZT=" + os.environ.get("VBX_MFCC_CZT", "default"), json.dumps(worst))
