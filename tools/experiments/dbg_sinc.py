import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import __graft_entry__ as g
pkg = g.load_package(); o = g.load_oracle()
vb = pkg.VoxBox(0)
N,H,SR=1200,480,48000.0
d = vb.synth_speech(6*48000, sample_offset=2*48000); audio = d.numpy()
for start in (1000, 2*48000+5000):
    x = audio[start:start+N]*o.window('hanning',N)
    r = o.normalize(o.autocorrelate(x,N))/o.window('hanning_lag',N)
    y = np.concatenate([r,np.zeros(N)])
    b=N//2; offset,nx=-b-1,2*b+1
    rng=np.random.default_rng(1)
    xs = rng.uniform(b+2+80, 2*b, 2000)
    for depth in (30,1200):
        got,st = vb.interpolate_sinc(y,offset,nx,xs,depth)
        exp = np.array([o.interpolate_sinc(y,offset,nx,v,depth)[1] for v in xs])
        err = np.abs(got-exp)
        print('start',start,'depth',depth,'max abs err',err.max(),'median',np.median(err),'scale',np.abs(exp).max())
    peaks=[k for k in range(1,b-1) if y[k-1]<y[k]>y[k+1] and 80<k<590]
    ix=np.array([k-offset+0.0 for k in peaks])
    got,st=vb.improve_extremum(y,offset,nx,ix,1200)
    dx=[];dy=[]
    for i,v in enumerate(ix):
        es,ex,ey=o.improve_extremum_sinc(y,offset,nx,v,1200)
        dx.append(got[i,0]-ex); dy.append(got[i,1]-ey)
    dx=np.array(dx);dy=np.array(dy)
    print(' npeaks',len(peaks),'max|dx|',np.abs(dx).max(),'max|dy|',np.abs(dy).max(), 'n(|dx|>1e-9)',(np.abs(dx)>1e-9).sum())
    print(' dx sample',dx[:8])
