# Simulation of the residual-evaluation plan of ns2d_fast.hip on float64 traces of the Jacobi solve (oracle/numpy_port.py):
# evaluations per solve of the proven plan (unweighted-norm bound) and of the extrapolated one, violations, log-convexity check.
import numpy as np, sys, math
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import numpy_port as NP
init=np.load(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))+'/tests/golden/rayleigh_128x64_init.npz')['fields']
e=NP.Rayleigh(init_fields=init,L=2.56,H=1.28)
acts=np.random.default_rng(1234).uniform(-1,1,(1,512,10))
a=NP.condition_actions(acts[0,0],e.C)
def poisson_trace(e,tol=1e-8):
    us,vs,phi=e.us,e.vs,e.phi; dx,dy,dt=e.dx,e.dy,e.dt
    b=((us[2:,1:-1]-us[1:-1,1:-1])/dx+(vs[1:-1,2:]-vs[1:-1,1:-1])/dy)/dt
    phi[:,:]=0; phin=np.zeros_like(phi); aW=[];a2=[]
    while True:
        phin[:,:]=phi
        phi[1:-1,1:-1]=0.5*((phin[2:,1:-1]+phin[:-2,1:-1])*dy*dy+(phin[1:-1,2:]+phin[1:-1,:-2])*dx*dx-b*dx*dx*dy*dy)/(dx*dx+dy*dy)
        phi[0,1:-1]=phi[1,1:-1];phi[-1,1:-1]=phi[-2,1:-1];phi[1:-1,-1]=phi[1:-1,-2];phi[1:-1,0]=phi[1:-1,1]
        d=phi-phin
        aW.append(float((d*d).sum())); a2.append(float((d[1:-1,1:-1]**2).sum()))
        if aW[-1]<=tol: break
    return np.array(aW),np.array(a2)
def plan(aW,a2,tol=1e-8,SAFE=1.25,JMAX=256):
    n=len(aW); k=0; nchk=0; kp=-1; lp=0; viol=0
    l2tol=math.log2(tol*SAFE)
    while True:
        nchk+=1
        if aW[k]<=tol: return nchk,k+1,viol
        l2a=math.log2(a2[k]); j=0
        if kp>=0:
            room=l2a-l2tol; rho=(l2a-lp)/(k-kp)
            if room>0: j=int(min(room/-rho,JMAX)) if rho<0 else JMAX
        lp=l2a;kp=k
        for i in range(1,j+1):
            if k+i<n and aW[k+i]<=tol: viol+=1
        k+=1+j
        if k>=n: return nchk,n,viol+100
tot=[0,0]
for it in range(12):
    e.bcs(a); e.predictor(); aW,a2=poisson_trace(e); e.p+=e.phi; e.corrector(); e.transport()
    r=[plan(aW,a2,SAFE=s) for s in (1.25,1.02)]
    print(it,len(aW),'a2/aW end %.3f'%(a2[-1]/aW[-1]),'decay bits/sweep first5 %.2f last5 %.3f'%(math.log2(aW[0]/aW[min(5,len(aW)-1)])/5,math.log2(aW[-6]/aW[-1])/5 if len(aW)>6 else 0),r, 'aW0/tol %.1e'%(aW[0]/1e-8))

def planW(aW,tol=1e-8,SAFE=1.003,JMAX=256,back=1):
    # heuristic: plan from the reference norm itself (ratio assumed non-decreasing), stop `back` sweeps early
    n=len(aW); k=0; nchk=0; kp=-1; lp=0; viol=0
    l2tol=math.log2(tol*SAFE)
    while True:
        nchk+=1
        if aW[k]<=tol: return nchk,k+1,viol
        l2a=math.log2(aW[k]); j=0
        if kp>=0:
            room=l2a-l2tol; rho=(l2a-lp)/(k-kp)
            if room>0: j=int(min(room/-rho,JMAX)) if rho<0 else JMAX
            j=max(0,j-back)
        lp=l2a;kp=k
        for i in range(1,j+1):
            if k+i<n and aW[k+i]<=tol: viol+=1
        k+=1+j
        if k>=n: return nchk,n,viol+100
print("heuristic on the reference norm")
e=NP.Rayleigh(init_fields=init,L=2.56,H=1.28)
tot=np.zeros(4); nv=0
for rep in range(3):
  e=NP.Rayleigh(init_fields=init,L=2.56,H=1.28)
  a=NP.condition_actions(acts[0,rep],e.C)
  for it in range(40):
    e.bcs(a); e.predictor(); aW,a2=poisson_trace(e); e.p+=e.phi; e.corrector(); e.transport()
    # log-convexity check of aW
    r=aW[1:]/aW[:-1]; bad=np.sum(r[1:]<r[:-1]*(1-1e-12))
    p1=plan(aW,a2,SAFE=1.02); p2=planW(aW,SAFE=1.003,back=1); p3=planW(aW,SAFE=1.0005,back=0)
    tot+= [len(aW),p1[0],p2[0],p3[0]]; nv+=p2[2]+p3[2]+p1[2]
    if it%10==0: print(rep,it,len(aW),'ratio decreases at',bad,'sweeps', p1,p2,p3)
print("sweeps, evals rigorous(1.02), evals heuristic(back 1), evals heuristic(back 0):",tot,"violations",nv)
