import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); import __graft_entry__ as g
pkg=g.load_package()
with pkg.VoxBox(0) as vb:
    Fd=1_000_000
    dd=vb.synth_speech(Fd*512); han=vb.window(pkg.WINDOW_HANNING,512)
    o_r=vb.empty((Fd,13)); o_a=vb.empty((Fd,13))
    for _ in range(3): vb.autocorr_lpc(dd,12,frame_len=512,stride=512,n_frames=Fd,window=han,out=(o_r,o_a))
    vb.sync(); t0=time.perf_counter()
    for _ in range(20): vb.autocorr_lpc(dd,12,frame_len=512,stride=512,n_frames=Fd,window=han,out=(o_r,o_a))
    vb.sync(); print(sys.argv[1] if len(sys.argv)>1 else "", "config2 ms", (time.perf_counter()-t0)/20*1e3, "listed", vb.last_lpc_exact_count())
