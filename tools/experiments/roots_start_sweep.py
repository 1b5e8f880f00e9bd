import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
pkg=g.load_package(); o=g.load_oracle()
synth=importlib.import_module(pkg.__name__+'.synth')
import roots_fast_model as R
N,H,F=512,512,30000
audio=synth.synth_speech((F-1)*H+N,3*48000)
s=o.soak(audio,N,H,0,F,12,48000.0,o.SOAK_FORMANTS)
A=np.concatenate([s["burg"], np.load('/root/repo/gpurun_out/redo_coeffs.npy')])
F=A.shape[0]
def run(start, fr_from=8):
    c=np.concatenate([A[:,::-1],np.ones((F,1))],axis=1); m=np.full(F,12); tot=np.zeros(F,int); flagged=np.zeros(F,bool)
    wave_its=0
    with np.errstate(all='ignore'):
        while np.any(m>2):
            top=int(m.max()); sel=np.nonzero(m>2)[0]; cc=c[sel]; n=m[sel].astype(float)
            x=np.full(sel.size,start[0]); y=np.full(sel.size,start[1]); done=np.zeros(sel.size,bool); its=np.zeros(sel.size,int)
            for it in range(32):
                p,dp,ddp,_=R.eval3(cc,top,x,y)
                G=dp/p; H=G*G-ddp/p; sq=np.sqrt((n-1)*(n*H-G*G)); d1,d2=G+sq,G-sq
                dz=n/np.where(np.abs(d1)>np.abs(d2),d1,d2)
                frac= it>=fr_from and (it-fr_from)%5==0
                if frac: dz=dz*R.FRACTIONS[((it-fr_from)//5)%7]
                upd=~done&(p!=0)
                x=np.where(upd,x-dz.real,x); y=np.where(upd,y-dz.imag,y); its+=upd
                done|=(p==0)|((not frac)&(np.abs(dz)<=1e-7*np.hypot(x,y)))
                if done.all(): break
            flagged[sel[~done]]=True; m[sel[~done]]=0
            tot[sel]+=its
            # wave-level cost: max its over groups of 64 consecutive frames
            wi=np.zeros(F,int); wi[sel]=its
            wave_its+=wi[:F//64*64].reshape(-1,64).max(axis=1).sum()
            ok=sel[done]; x,y,cc=x[done],np.abs(y[done]),cc[done]
            real=np.abs(y)<=1e-9*np.hypot(x,y)
            if np.any(~real):
                _,_,_,b=R.eval3(cc[~real],top,x[~real],y[~real]); nb=np.zeros((int(np.sum(~real)),13)); nb[:,:top-1]=b[:,2:top+1]
                c[ok[~real]]=nb; m[ok[~real]]-=2
            if np.any(real):
                q=np.zeros((int(np.sum(real)),13)); t=np.zeros(int(np.sum(real)))
                for k in range(12,-1,-1):
                    ck=cc[real][:,k]; q[:,k]=t; t=x[real]*t+ck
                c[ok[real]]=q; m[ok[real]]-=1
    print("start",start,"fr_from",fr_from,": flagged",flagged.sum(),"mean its/frame %.1f"%tot.mean(),"wave-level its per wave %.1f"%(wave_its/(F//64)))
for st in ((0.95,0.2),(0.8,0.4),(0.85,0.45),(1.0,0.1),(0.7,0.2),(0.9,0.1),(0.8,0.6),(1.1,0.3)):
    run(st)
