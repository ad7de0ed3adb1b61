"""Numerical experiment behind the F(4x4,3x3) interpolation points of csrc/winograd.hip: fp32 simulation of the forward and
weight-gradient Winograd pipelines (256-channel / 1024-tile reductions, N(0,1) data) for several point sets against float64."""
import numpy as np
from fractions import Fraction as F
def mats(pts):
    n=len(pts)+1; m=4; r=3
    AT=np.zeros((m,n)); G=np.zeros((n,r))
    for j,a in enumerate(pts):
        for i in range(m): AT[i,j]=a**i
        N=np.prod([a-b for k,b in enumerate(pts) if k!=j])
        for l in range(r): G[j,l]=a**l/N
    AT[m-1,n-1]=1; G[n-1,r-1]=1
    # solve BT
    rows=[];rhs=[]
    for i in range(m):
        for l in range(r):
            for k in range(n):
                row=np.zeros((n,n))
                for j in range(n): row[j,k]=AT[i,j]*G[j,l]
                rows.append(row.ravel()); rhs.append(1.0 if i+l==k else 0.0)
    BT=np.linalg.lstsq(np.array(rows),np.array(rhs),rcond=None)[0].reshape(n,n)
    return AT,G,BT
def f32(x): return x.astype(np.float32)
def run(pts,C=256,T=64,seed=0, scaleG=None):
    AT,G,BT=mats(pts)
    rng=np.random.default_rng(seed)
    d=rng.standard_normal((T,C,6,6)); g=rng.standard_normal((C,3,3))*0.05
    ref=np.zeros((T,4,4))
    for i in range(4):
        for j in range(4):
            ref[:,i,j]=np.einsum('tcab,cab->t',d[:,:,i:i+3,j:j+3],g)
    ATf,Gf,BTf=f32(AT),f32(G),f32(BT)
    U=f32(np.einsum('ia,cab->cib',Gf,f32(g))); U=f32(np.einsum('cib,jb->cij',U,Gf))
    V=f32(np.einsum('ia,tcab->tcib',BTf,f32(d))); V=f32(np.einsum('tcib,jb->tcij',V,BTf))
    # channel reduction in fp32 sequentially (fmaf-chain like)
    M=np.zeros((T,6,6),np.float32)
    for c in range(C): M=(M+V[:,c]*U[c]).astype(np.float32)
    Y=f32(np.einsum('ia,tab->tib',ATf,M)); Y=f32(np.einsum('tib,jb->tij',Y,ATf))
    err=np.abs(Y-ref).max()/np.abs(ref).max()
    rms=np.sqrt(((Y-ref)**2).mean())/np.sqrt((ref**2).mean())
    return err,rms,np.abs(BT).max(),np.abs(AT).max(),np.abs(G).max()
def direct(C=256,T=64,seed=0):
    rng=np.random.default_rng(seed)
    d=rng.standard_normal((T,C,6,6)); g=rng.standard_normal((C,3,3))*0.05
    ref=np.zeros((T,4,4)); y=np.zeros((T,4,4),np.float32)
    for i in range(4):
        for j in range(4):
            ref[:,i,j]=np.einsum('tcab,cab->t',d[:,:,i:i+3,j:j+3],g)
            acc=np.zeros(T,np.float32)
            for c in range(C):
                for a in range(3):
                    for b in range(3):
                        acc=(acc+f32(d[:,c,i+a,j+b])*np.float32(g[c,a,b])).astype(np.float32)
            y[:,i,j]=acc
    return np.abs(y-ref).max()/np.abs(ref).max(), np.sqrt(((y-ref)**2).mean())/np.sqrt((ref**2).mean())
def run_w(pts,T=1024,seed=0):
    AT,G,BT=mats(pts)
    rng=np.random.default_rng(seed)
    d=rng.standard_normal((T,6,6)); dy=rng.standard_normal((T,4,4))
    ref=np.zeros((3,3))
    for a in range(3):
        for b in range(3):
            ref[a,b]=(dy*d[:,a:a+4,b:b+4]).sum()
    ATf,Gf,BTf=f32(AT),f32(G),f32(BT)
    V=f32(np.einsum('ia,tab->tib',BTf,f32(d))); V=f32(np.einsum('tib,jb->tij',V,BTf))
    dM=f32(np.einsum('ai,tab->tib',ATf,f32(dy))); dM=f32(np.einsum('tib,bj->tij',dM,ATf))
    dU=np.zeros((6,6),np.float32)
    for t in range(T): dU=(dU+dM[t]*V[t]).astype(np.float32)
    dw=f32(np.einsum('ia,ij->aj',Gf,dU)); dw=f32(np.einsum('aj,jb->ab',dw,Gf))
    return np.sqrt(((dw-ref)**2).mean())/np.sqrt((ref**2).mean())
def direct_w(T=1024,seed=0):
    rng=np.random.default_rng(seed)
    d=rng.standard_normal((T,6,6)); dy=rng.standard_normal((T,4,4))
    ref=np.zeros((3,3)); out=np.zeros((3,3),np.float32)
    for a in range(3):
        for b in range(3):
            ref[a,b]=(dy*d[:,a:a+4,b:b+4]).sum()
            acc=np.float32(0)
            for t in range(T):
                for i in range(4):
                    for j in range(4): acc=np.float32(acc+np.float32(dy[t,i,j])*np.float32(d[t,a+i,b+j]))
            out[a,b]=acc
    return np.sqrt(((out-ref)**2).mean())/np.sqrt((ref**2).mean())
print("direct wgrad", np.mean([direct_w(seed=s) for s in range(3)]))
for pts in ([0,1,-1,2,-2],[0,1,-1,2,-0.5],[0,0.5,-0.5,-2,1.5],[0,0.5,-0.5,1.5,-1.5],[0,1,-1,0.5,-2],[0,1,-1,0.5,-0.5]):
    ef=np.mean([run(pts,seed=s,T=64)[1] for s in range(6)])
    ew=np.mean([run_w(pts,seed=s) for s in range(6)])
    print(pts,"fwd rms %.2e  wgrad rms %.2e"%(ef,ew))
