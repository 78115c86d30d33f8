"""cfg3-size forward+backward, three handles x two runs: predictions and gradients must be identical bit for bit
(persistent kernels, split-K folds and consumer gathers all sum in a fixed order).
usage: python tools/determinism.py [C] [nContractions]      (defaults 64 18; 10 10 / 10 50: the SMP_2D_ver6 / ver7 wirings on the fused level)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from inputs import smp_params, synthetic_molecule
from graphflow_amd.smp import SMPOmega
L,C,F,D,cap=3,(int(sys.argv[1]) if len(sys.argv)>1 else 64),5,5,29
nK=int(sys.argv[2]) if len(sys.argv)>2 else 18
mols=[]; tg=[]
for i in range(1024):
    a,f,t=synthetic_molecule(i); mols.append((a,f)); tg.append(t)
n0=SMPOmega(L,C,F,D,cap,True,nContractions=nK,custom_matmul=(nK!=18)); npar=n0.n_params; n0.close()
p=torch.as_tensor((smp_params(C,F,D,L,1) if nK==18 else np.random.default_rng(1).uniform(-1,1,npar)/np.sqrt(nK*C)).astype(np.float32)).cuda(); t=torch.as_tensor(np.array(tg,dtype=np.float32)).cuda()
outs=[]
for rep in range(3):
    net=SMPOmega(L,C,F,D,cap,True,nContractions=nK,custom_matmul=(nK!=18)); net.prepare(mols)
    for k in range(2):
        pred,loss,feat=net.forward(p,t); g=torch.empty(net.n_params,device='cuda'); net.backward(p,g)
        outs.append((pred.clone(),g.clone()))
torch.cuda.synchronize()
ok=all(torch.equal(outs[0][0],o[0]) and torch.equal(outs[0][1],o[1]) for o in outs[1:])
print("bit-reproducible across", len(outs), "runs / 3 handles:", ok, float(outs[0][1].abs().max()))
assert ok
