"""The one-pass Burg guard (csrc/vbx_burg_fast.hpp) against the ACTUAL error of the one-pass recursion, on real speech: frames of
tests/golden/sample-two_vowels.wav (44.1 kHz, order 13) through the numpy model of the kernel (tests/burg_one_pass_model.py)
and through the direct recursion in long double.  CPU only.  Round 5: of the 166 / 243 frames the guard
turns away at 1024 / 512, 33 are off by more than 1e-7 and 11 by more than the gate of 5e-7 -- the error is real, and no
cheaper test than the direct recursion itself tells those frames from the rest (every rigorous bound is worst-case in the lag
sums' rounding, typical errors are 20x below it)."""
import os, sys, wave, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from burg_one_pass_model import burg_one_pass, parity_metric, KAPPA_EPS, TARGET
def readwav(p):
    w=wave.open(p); n=w.getnframes(); d=np.frombuffer(w.readframes(n),dtype='<i2'); ch=w.getnchannels(); sr=w.getframerate()
    return d.reshape(-1,ch)[:,0].astype(np.float64)/32767.0, sr
def burg_direct(x,P,dt=np.longdouble):
    x=x.astype(dt); N=len(x)
    b1=x[:-1].copy(); b2=x[1:].copy()   # f=b2 , b=b1  (conceptually)
    a=np.zeros(P,dtype=dt)
    # standard memcof
    wk1=x[:-1].copy(); wk2=x[1:].copy(); d=np.zeros(P,dtype=dt); wkm=np.zeros(P,dtype=dt)
    for k in range(P):
        m=N-k-1
        num=np.sum(wk1[:m]*wk2[:m]); den=np.sum(wk1[:m]**2+wk2[:m]**2)
        d[k]=2*num/den
        for i in range(k): d[i]=wkm[i]-d[k]*wkm[k-1-i]
        if k==P-1: break
        wkm[:k+1]=d[:k+1]
        n1=wk1[:m-1]-wkm[k]*wk2[:m-1]; n2=wk2[1:m]-wkm[k]*wk1[1:m]
        wk1[:m-1]=n1; wk2[:m-1]=n2
    return -d   # sign? compare up to sign
x,sr=readwav(os.path.join(ROOT, 'tests', 'golden', 'sample-two_vowels.wav'))
print(sr,len(x))
for N,H,P in ((1024,512,13),(2048,1024,13),(1103,441,13)):
    F=(len(x)-N)//H+1
    w=0.5*(1-np.cos(2*np.pi*np.arange(N)/N))
    X=np.stack([x[t*H:t*H+N]*w for t in range(F)])
    co,tr=burg_one_pass(X,P)
    ref=np.stack([burg_direct(X[t],P).astype(np.float64) for t in range(F)])
    ref64=np.stack([burg_direct(X[t],P,np.float64) for t in range(F)])
    if np.abs(co+ref).max()<np.abs(co-ref).max(): ref=-ref; ref64=-ref64
    m=parity_metric(co,ref); m64=parity_metric(ref64,ref)
    print(N,H,P,'frames',F,'trusted',tr.sum(),'rejected',(~tr).sum())
    print(' actual err one-pass: max',m.max(),'median',np.median(m),' direct f64 err max',m64.max())
    print(' rejected frames: actual err quantiles',np.quantile(m[~tr],[0.5,0.9,0.99,1.0]) if (~tr).any() else None)
    print(' frames with actual err>5e-7:',(m>5e-7).sum(),' >1e-7:',(m>1e-7).sum(), '>1e-8',(m>1e-8).sum())
