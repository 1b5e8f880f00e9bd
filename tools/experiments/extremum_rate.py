# rate of improve_extremum at 16 lanes per point, 4 points per wave (the engine a multi-frame refine kernel would use)
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0); o = g.load_oracle()
import importlib
synth = importlib.import_module(g.PKG_NAME + ".synth")
N,H,SR=1200,480,48000.0
audio = synth.synth_speech(10*48000+N, sample_offset=0)
w = o.window("hanning", N); lw = o.window("hanning_lag", N)
x = audio[100*H:100*H+N]*w
r = o.autocorrelate(x, N); r = r/np.max(np.abs(r)); yv = r/lw
y = np.concatenate([yv, np.zeros(N)])
b = N//2; offset = -b-1; nx = b-offset
k = 1 + int(np.argmax(yv[80:600])) + 79
print("peak lag", k, "y", yv[k])
M = 1<<20
rng = np.random.default_rng(0)
ix = (k - offset) + rng.uniform(-0.3, 0.3, M)
dy = vb.to_device(y); dix = vb.to_device(ix); out = vb.empty((M,2)); st = vb.empty(M, np.int32)
for i in range(3):
    vb.timer_begin()
    vb._check(vb.L.vbx_improve_extremum_f64(vb.ctx, dy.ptr, y.size, offset, nx, dix.ptr, M, 1200, out.ptr, st.ptr))
    ms = vb.timer_end()
print(f"{M} refinements in {ms:.2f} ms -> {M/ms/1e3:.1f} M refinements/s")
