import os as _os, sys as _sys
_HERE = _os.path.dirname(_os.path.abspath(__file__))
_sys.path.insert(0, _HERE); _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_HERE)))
import numpy as np
from jacobi_emulation import make_mats, pi_perm
def jacobi_m(G, W, m, tol2=1e-14, big2=1e-6, frac=0.05, maxs=20):
    n=G.shape[0]; G=G.copy(); V=np.eye(n); tr=np.trace(G)
    pi=pi_perm(n); inv=np.argsort(pi)
    U,S,Vt=np.linalg.svd(W,full_matrices=False); best=(U[:,:m]*S[:m])@Vt[:m]
    for s in range(maxs):
        lam_m=np.sort(np.diag(G))[::-1][m-1]
        fl=(frac*lam_m)**2
        anyr=big=False
        for r in range(n-1):
            J=np.eye(n)
            for k in range(n//2):
                a,b_,g=G[2*k,2*k],G[2*k+1,2*k+1],G[2*k,2*k+1]
                sc=max(abs(a*b_),fl); g2=g*g
                if not g2>max(tol2*sc,(1e-15*tr)**2): continue
                anyr=True
                if g2>big2*sc: big=True
                d=b_-a; h=np.sqrt(d*d+4*g*g); t=2*g/(d+np.copysign(h,d))
                c=1/np.sqrt(1+t*t); sn=c*t
                J[2*k,2*k]=c; J[2*k,2*k+1]=sn; J[2*k+1,2*k]=-sn; J[2*k+1,2*k+1]=c
            G=J.T@G@J; V=V@J
            G=G[np.ix_(inv,inv)]; V=V[:,inv]
        if not anyr or not big: break
    lam=np.diag(G); o=np.argsort(-lam)[:m]; Q=V[:,o]
    prod=Q@(Q.T@W)
    perr=np.abs(prod-best).max()/np.abs(W).max()
    serr=np.abs(np.sqrt(np.maximum(lam[o],0))-S[:m]).max()/S[0]
    srel=(np.abs(np.sqrt(np.maximum(lam[o],0))-S[:m])/S[:m]).max()
    return s+1,perr,serr,srel
mats=make_mats()
for kw in [dict(),dict(big2=1e-4),dict(big2=1e-4,frac=0.2),dict(tol2=1e-12,big2=1e-4,frac=0.2),dict(tol2=1e-12,big2=1e-3,frac=0.3)]:
    r=[]
    for Bm in mats[::2]:
        W=Bm if Bm.shape[0]<=Bm.shape[1] else Bm.T
        r.append(jacobi_m(W@W.T,W,20,**kw))
    r=np.array(r)
    print(kw,'sweeps mean %.2f max %d | prod err max %.1e | sigma abs err %.1e rel %.1e'%(r[:,0].mean(),r[:,0].max(),r[:,1].max(),r[:,2].max(),r[:,3].max()))
