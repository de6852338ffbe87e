import json, os, sys, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT)
from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, cases
fx=json.load(open(os.path.join(ROOT,"tests","golden","fixtures.json")))
rng=np.random.Generator(np.random.PCG64([1234,0])); B=65536
ts=cases.case2_tsteps(); u0=cases.case2_u0(B,rng)
p=np.array(json.load(open(os.path.join(ROOT,"tests","golden","case2_hard_p.json")))["p"])
for lanes in (1,2):
    node=NeuralODE(ODEProblem(PRESET_CASE2,ts)); node.set_ensemble(u0,np.zeros((B,6,len(ts))),np.ones(6)); node.set_lanes_per_traj(lanes)
    k=[]
    for _ in range(6):
        node.loss_and_grad(p); k.append(node.stats()["kernel_ms"])
    na,nr=node.step_counts(); n=(na+nr).astype(np.int64)
    q=np.percentile(n,[0,10,50,90,99,99.9,100])
    print("lanes",lanes,"kernel_ms",np.median(k),"attempts mean %.1f"%n.mean(),"pct 0/10/50/90/99/99.9/100", "/".join("%d"%v for v in q), "acc mean %.1f rej mean %.1f"%(na.mean(),nr.mean()))
    node.close()
