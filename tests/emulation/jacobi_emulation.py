import os as _os, sys as _sys
_HERE = _os.path.dirname(_os.path.abspath(__file__))
_sys.path.insert(0, _HERE); _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_HERE)))
import numpy as np, sys
from oracle import mps_oracle as mo

def make_mats(N=24,M=20,b=1500,L=2,seed=0):
    rng=np.random.default_rng(seed); D=2
    p=rng.random((b,N))*(rng.random((b,N))>0.81)
    X=np.stack([np.sin(np.pi*p/2),np.cos(np.pi*p/2)],-1)
    y=rng.integers(0,L,b)
    st=mo.MPSState(N,D,L,M,mo.random_cores(N,M,D,L,rng=rng,scale=M*0.64))
    mo.calibrate(st,X)
    mats=[]
    y1h=mo.one_hot(y,L)
    for sw in range(2):
        f=mo.forward(st,X); left=st.l_pos==N-1
        if left: st.Renv={}
        else: st.Lenv={}
        for j in range(N-1):
            rec={}
            f=mo.sweep_step(st,f,y1h,1e-3,1e-3,True,left,'softmax','full_cross_ent',0.1,'fixed',record=rec)
            Bm=rec['Bmat'].astype(np.float32).astype(np.float64)
            if min(Bm.shape)==2*M: mats.append(Bm)
    return mats

def pi_perm(ne):
    np_=ne//2; pi=np.zeros(ne,int)
    for pos in range(ne):
        k=pos>>1
        if pos&1: pi[pos]= (2 if np_>1 else 1) if k==0 else 2*(k-1)+1
        else: pi[pos]= 0 if k==0 else (2*k+1 if k==np_-1 else 2*(k+1))
    return pi

def jacobi(G, tol2=1e-22, small2=1e-12, absf=1e-15, presort=False, maxs=30, f32=False):
    n=G.shape[0]; G=G.copy(); tr=np.trace(G); floor2=(absf*tr)**2
    if presort:
        o=np.argsort(-np.diag(G)); G=G[np.ix_(o,o)]
    pi=pi_perm(n); inv=np.argsort(pi)
    sweeps=0; offs=[]
    for s in range(maxs):
        anyr=big=False
        for r in range(n-1):
            J=np.eye(n)
            for k in range(n//2):
                a,b_,g=G[2*k,2*k],G[2*k+1,2*k+1],G[2*k,2*k+1]
                g2=g*g; ab=abs(a*b_)
                if not g2>max(tol2*ab,floor2): continue
                anyr=True
                if g2>max(small2*ab,1e6*floor2): big=True
                d=b_-a; h=np.sqrt(d*d+4*g*g); t=2*g/(d+np.copysign(h,d))
                if f32: t=float(np.float32(t))
                c=1/np.sqrt(1+t*t); sn=c*t
                J[2*k,2*k]=c; J[2*k,2*k+1]=sn; J[2*k+1,2*k]=-sn; J[2*k+1,2*k+1]=c
            G=J.T@G@J
            G=G[np.ix_(inv,inv)]   # new[pi[i]] = old[i]
        sweeps+=1
        off=np.sqrt((G**2).sum()-(np.diag(G)**2).sum())/tr; offs.append(off)
        if not anyr or not big: break
    return sweeps, offs, np.sort(np.diag(G))[::-1]

if __name__=='__main__':
    mats=make_mats()
    print(len(mats),'matrices', mats[0].shape)
    for name,kw in [('base',{}),('presort',dict(presort=True)),('small 1e-8',dict(small2=1e-8)),('small1e-8+presort',dict(small2=1e-8,presort=True)),('f32 t',dict(f32=True))]:
        res=[]; err=[]
        for Bm in mats[::3]:
            G=Bm@Bm.T if Bm.shape[0]<=Bm.shape[1] else Bm.T@Bm
            s,offs,lam=jacobi(G,**kw)
            ref=np.linalg.svd(Bm,compute_uv=False)**2
            res.append(s); err.append(np.abs(np.sqrt(np.maximum(lam,0))-np.sqrt(ref)).max()/np.sqrt(ref[0]))
        print(name,'sweeps mean %.2f max %d  sigma err %.1e'%(np.mean(res),max(res),max(err)), 'offs of last:',['%.0e'%o for o in offs])
