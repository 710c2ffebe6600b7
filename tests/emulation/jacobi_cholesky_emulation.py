"""CPU emulation: one pivoted-Cholesky step (G = L L^T, G' = L^T L) before the Jacobi iteration -- sweeps saved per training
pass and their relation to off(G)/trace(G), which the kernel uses to decide whether to take the step."""
import os as _os, sys as _sys
_HERE = _os.path.dirname(_os.path.abspath(__file__))
_sys.path.insert(0, _HERE); _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_HERE)))

import sys, numpy as np
import jacobi_warm_start_emulation as J
from oracle import mps_oracle as mo
def pchol(G):
    n=G.shape[0]; A=G.copy(); perm=np.arange(n); L=np.zeros((n,n))
    for k in range(n):
        j=k+np.argmax(np.diag(A)[k:])
        if j!=k:
            A[[k,j]]=A[[j,k]]; A[:,[k,j]]=A[:,[j,k]]; L[[k,j]]=L[[j,k]]; perm[[k,j]]=perm[[j,k]]
        piv=A[k,k]
        if piv<=1e-30*np.trace(G): L[k,k]=0; continue
        L[k,k]=np.sqrt(piv); L[k+1:,k]=A[k+1:,k]/L[k,k]
        A[k+1:,k+1:]-=np.outer(L[k+1:,k],L[k+1:,k])
    return L,perm
# collect matrices per pass with the harness (device ordering, float32-rounded cores)
class H2(J.Harness):
    def svd(self,Bmat,m):
        Bm=Bmat.astype(np.float32).astype(np.float64)
        W=Bm if Bm.shape[0]<=Bm.shape[1] else Bm.T
        n=W.shape[0]; nb=n//2
        perm=np.array([d_*nb+a for a in range(nb) for d_ in range(2)]) if n%2==0 else np.arange(n)
        if n==40: self.mats.append((self.pass_no,W[perm].copy(),m))
        return super().svd(Bmat,m)
N,M,b,L,D=24,20,1500,2,2
rng=np.random.default_rng(0)
p=rng.random((b,N))*(rng.random((b,N))>0.81)
X=np.stack([np.sin(np.pi*p/2),np.cos(np.pi*p/2)],-1); y=rng.integers(0,L,b)
st=mo.MPSState(N,D,L,M,mo.random_cores(N,M,D,L,rng=rng,scale=M*0.64)); mo.calibrate(st,X); y1h=mo.one_hot(y,L)
H=H2(False,1e-6); H.mats=[]; orig=mo.tensor_svd; mo.tensor_svd=H.svd
for sw in range(8):
    H.pass_no=sw
    f=mo.forward(st,X); left=st.l_pos==N-1
    if left: st.Renv={}
    else: st.Lenv={}
    for j in range(N-1):
        H.key=(st.l_pos,left)
        f=mo.sweep_step(st,f,y1h,1e-3,1e-3,True,left,'softmax','full_cross_ent',0.1,'fixed')
mo.tensor_svd=orig
for ps in (0,1,3,7):
    base=[];pc=[]
    for (q,W,m) in H.mats:
        if q!=ps: continue
        G=W@W.T
        s0,_,_=J.jacobi(G,m,BIG2=1e-6)
        Lc,perm=pchol(G); G1=Lc.T@Lc
        s1,_,_=J.jacobi(G1,m,BIG2=1e-6)
        base.append(s0);pc.append(s1)
    print('pass',ps,'plain %.2f'%np.mean(base),'after one pivoted Cholesky step %.2f'%np.mean(pc),base,pc)
print()
for ps in range(8):
    offs=[]; gains=[]
    for (q,W,m) in H.mats:
        if q!=ps: continue
        G=W@W.T
        off=np.sqrt((G**2).sum()-(np.diag(G)**2).sum())/np.trace(G)
        # alternative cheap predictor: relative size of the largest off-diagonal of the first round's 2x2 blocks etc.
        s0,_,_=J.jacobi(G,m,BIG2=1e-6); Lc,perm=pchol(G); s1,_,_=J.jacobi(Lc.T@Lc,m,BIG2=1e-6)
        offs.append(off); gains.append(s0-s1)
    print('pass',ps,'off/trace',' '.join('%.2f'%o for o in offs)); print('      sweeps saved',gains)
