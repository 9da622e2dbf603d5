"""CPU emulation of the engine's forward-weight rounding on top of the float64 oracle ("Saturated weights"):
how much second-order error do 1-ulp errors in the saved weights cause, and does forming the largest weight as 1 - others remove it?
usage: python tools/emu_weights.py   (CPU only, ~1 minute)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, datagen
from oracle import oracle
f32=np.float32
def ref(theta,A,Z,variant,dt):
    th=theta.astype(dt); a=A.astype(dt)
    Vt,E,Q,Ef = oracle.fwd_bwd(th,a,None,variant,omp=False)
    Ed,Vtd,Qd = oracle.double_backward(Q,Ef,Z.astype(dt),None,omp=False)
    return dict(Vt=Vt,E=E,Ed=Ed,Vtd=Vtd,Q=Q,Ef=Ef,Qd=Qd)
def fwdV(theta,A,variant):
    N,M=theta.shape; V=np.zeros((N+1,M+1)); lo=2 if variant else 1
    for i in range(lo,N+1):
        for j in range(lo,M+1):
            a=float(A[i-1,j-1]); X=np.array([a+V[i-1,j],V[i-1,j-1],a+V[i,j-1]]); m=X.max()
            V[i,j]=float(theta[i-1,j-1])+m+np.log(np.exp(X-m).sum())
    return V
def ulp(rng): return 1.0+rng.uniform(-1,1)*6e-8
def eng_weights(V,A,variant,fix,rng,vnoise=0.0,newton=False):
    N,M=A.shape; qx=np.zeros((N+2,M+2)); qy=np.zeros((N+2,M+2)); lo=2 if variant else 1
    Vn=V+vnoise*rng.standard_normal(V.shape) if vnoise>0 else V
    for i in range(lo,N+1):
        for j in range(lo,M+1):
            a=float(A[i-1,j-1]); vu,vd,vl=Vn[i-1,j],Vn[i-1,j-1],Vn[i,j-1]
            R=max(vu,vd,vl)
            u=f32(np.exp(vu-R)); d=f32(np.exp(vd-R)); x=f32(np.exp(vl-R))
            ca=f32(np.exp(a)*ulp(rng))
            ssum=f32(np.float64(ca)*np.float64(f32(u+x))+np.float64(d))
            if ssum==0: continue
            r=f32(1.0/np.float64(ssum)) if newton else f32(1.0/np.float64(ssum)*ulp(rng))
            tq=f32(ca*r); wx=f32(tq*u); wy=f32(tq*x)
            if fix:
                wm=f32(d*r)
                if wx>=wy and wx>=wm: wx=f32(f32(1)-f32(wy+wm))
                elif wy>=wm: wy=f32(f32(1)-f32(wx+wm))
            qx[i,j]=wx; qy[i,j]=wy
    return qx,qy
def adj(qx,qy,Ef32,Z):
    N=qx.shape[0]-2; M=qx.shape[1]-2
    qm=(1.0-qx)-qy; qm[0,:]=qm[-1,:]=0; qm[:,0]=qm[:,-1]=0
    # cells never computed (SW border) have q=0 -> qm must be 0 there too
    qm[(qx==0)&(qy==0)]=np.where(True,qm[(qx==0)&(qy==0)],0)
    Vd=np.zeros((N+1,M+1)); Qd=np.zeros((N+2,M+2,3))
    for i in range(1,N+1):
        for j in range(1,M+1):
            a0=Vd[i-1,j]; a1=Vd[i-1,j-1]; a2=Vd[i,j-1]
            tot=qy[i,j]*a2+(qm[i,j]*a1+qx[i,j]*a0)
            Vd[i,j]=Z[i-1,j-1]+tot
            Qd[i,j,0]=f32(qx[i,j]*(a0-tot)); Qd[i,j,2]=f32(qy[i,j]*(a2-tot)); Qd[i,j,1]=-(Qd[i,j,0]+Qd[i,j,2])
    Ed=np.zeros((N+2,M+2)); E=Ef32.astype(np.float64)
    for i in range(N,0,-1):
        for j in range(M,0,-1):
            Ed[i,j]=(Qd[i+1,j,0]*E[i+1,j]+qx[i+1,j]*Ed[i+1,j]+Qd[i+1,j+1,1]*E[i+1,j+1]+qm[i+1,j+1]*Ed[i+1,j+1]
                     +Qd[i,j+1,2]*E[i,j+1]+qy[i,j+1]*Ed[i,j+1])
    return Ed[1:-1,1:-1].astype(f32)
rng=np.random.default_rng(1)
cases=[(2,688,1,5.0,1.0,0.0,1),(3,1500,0,8.0,1.0,0.0,2),(5,2000,0,30.0,10.0,0.5,3),(1200,3,0,8.0,0.0,0.0,4),(2,1000,0,1.0,1.0,0.0,5),(6,1900,0,30.0,0.0,0.0,7),(7,2048,0,8.0,10.0,0.0,8),(40,300,0,1.0,1.0,0.0,9)]
for (N,M,variant,ts,as_,ao,seed) in cases:
    theta,A=datagen.theta_A(900+seed,1,N,M); theta=(theta*ts).astype(f32); A=(A*as_+ao).astype(f32)
    Z=datagen.normal(950+seed,(1,N,M))
    r32=ref(theta,A,Z,variant,f32)
    if variant==1: continue_sw=True
    V=fwdV(theta[0],A[0],variant)
    sc=max(1.0,np.abs(r32["Ed"]).max())
    out=f"{(N,M,variant,ts,as_,ao)} scale={sc:.3g}"
    for fix in (False,True):
      for newton in (False,):
        for vn in (3e-7,3e-6):
            qx,qy=eng_weights(V,A[0],variant,fix,rng,vn,newton)
            # SW: row/col 1 cells: weights zero, qm must be zero => handled: qm=1 there! fix:
            Ed=adj(qx,qy,r32["Ef"][0],Z[0].astype(np.float64)) if variant==0 else None
            if Ed is None: continue
            out+=f" | fix={int(fix)} nw={int(newton)} vn={vn:g}: {np.abs(Ed-r32['Ed'][0]).max()/sc:.2e}"
    print(out,flush=True)
