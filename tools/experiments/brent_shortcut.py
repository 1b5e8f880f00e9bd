import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import __graft_entry__ as g
pkg=g.load_package(); o=g.load_oracle()
synth=importlib.import_module(pkg.__name__+'.synth')
N=1200; H=480; sr=48000.0; fmin,fmax=75.0,600.0
def candidates(frame):
    wl=o.window("hanning_lag",N)
    r=o.autocorrelate(frame,N); r=o.normalize(r); y=np.concatenate([r/wl, np.zeros(N)])
    bix=N//2; offset=-bix-1; nx=bix-offset
    out=[]
    for k in range(1,bix-1):
        if not (y[k-1]<y[k] and y[k+1]<y[k]): continue
        dr=0.5*(y[k+1]-y[k-1]); d2r=2*y[k]-(y[k-1]-y[k+1])
        freq=sr/(k+dr/d2r)
        if not (freq>fmin and freq<fmax): continue
        n2=sr/freq-offset
        st,xm,ym=o.improve_extremum_sinc(y,offset,nx,n2,1200)
        xm+=offset
        if ym>1: ym=1/ym
        # shortcut prediction
        lo=min(y[k-1],y[k+1]); sp=lo if lo<=1 else 1/lo
        out.append((k, sr/xm, ym, sr/k, sp, y[k-1], y[k], y[k+1], dr/d2r, xm))
    return out
if __name__=='__main__':
    kind=sys.argv[1]; F=int(sys.argv[2])
    rng=np.random.default_rng(1)
    w=o.window("hanning",N)
    if kind=='speech':
        audio=synth.synth_speech((F-1)*H*3+N, 0)
        frames=[audio[t*H*3:t*H*3+N]*w for t in range(F)]
    elif kind=='noise':
        frames=[rng.standard_normal(N)*w for t in range(F)]
    elif kind=='tones':
        t=np.arange(N)
        frames=[(np.sin(2*np.pi*rng.uniform(80,500)/sr*t+rng.uniform(0,6))+rng.uniform(0,0.5)*np.sin(2*np.pi*rng.uniform(80,2000)/sr*t)+10**-rng.uniform(1,4)*rng.standard_normal(N))*w for _ in range(F)]
    tot=0; bad=0; topbad=0; ntop=0; worst=[]
    for fr in frames:
        c=candidates(fr)
        if not c: continue
        c=np.array(c)
        okf=np.abs(c[:,1]-c[:,3])<=1e-4*np.abs(c[:,1]); oks=np.abs(c[:,2]-c[:,4])<=1e-4
        tot+=len(c); b=~(okf&oks); bad+=b.sum()
        top=np.argmax(c[:,2]); ntop+=1
        if c[top,2]>0.2:
            if b[top]: topbad+=1; worst.append(c[top])
    print(kind,"candidates",tot,"prediction off by >1e-4:",bad,"(%.3f%%)"%(100*bad/max(tot,1)),"; voiced top candidates",ntop,"off:",topbad)
    for wv in worst[:8]: print("   k=%d ref f=%.4f s=%.6f  pred f=%.4f s=%.6f  y[k-1..k+1]=%.5f %.5f %.5f d0=%.2e xm=%.6f"%tuple(wv))
