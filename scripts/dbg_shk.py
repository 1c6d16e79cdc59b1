import sys, random; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'.')
import numpy as np, torch
from oracle import oracle as O
from beacon_amd import vec as V
g=np.load('tests/golden/shkadov.npz')
init=np.stack([g['j5_h_init'],g['j5_q_init']])
for B in (1,2):
    e=O.shkadov(n_jets=5, init_fields=init); e.rand_init=False; e.reset()
    env=V.VecShkadov(B,"cuda:0","f64",init,n_jets=5); env.reset()
    np.random.seed(9)
    for i in range(125):
        nz=np.random.uniform(-e.sigma,e.sigma,50)
        o,_,_,_,_=e.step(None,nz)
        obs,_,_,_,_=env.step(None, np.tile(nz,(B,1)))
        st=env.get_state().cpu().numpy()[0]
        d=np.abs(st[0]-e.h).max()
        if i<5 or i%10==0 or d>1e-9: print(B,i,d, np.abs(obs[0].cpu().numpy()-o).max(), np.argmax(np.abs(st[0]-e.h)))
        if d>1e-6: break
