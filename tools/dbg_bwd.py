import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests/golden'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import graphflow_amd as gf
from graphflow_amd import _lib
from oracle import pyoracle as po
from inputs import adjacency
from util import rel_err
o = po.oracle()
rng = np.random.default_rng(0)
def dev(x): return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()
for (N,C) in ((3,2),(5,3),(4,4),(8,64)):
  A = adjacency("weighted", N, rng)
  for generic in (1,0):
    _lib.load().gf_debug_force_generic(generic)
    errs=[]
    for k in range(18):
      G = np.zeros((N,N,18,C)); G[:,:,k,:] = rng.uniform(0,1,(N,N,C))
      ref = o.contract_backward(18, G, A)
      got = gf.contract_backward(dev(G[None]), dev(A[None]), 18).cpu().numpy()[0]
      errs.append(round(rel_err(got, ref),6))
    print(N,C,'generic' if generic else 'fast', errs)
