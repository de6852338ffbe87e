import json, os, sys, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT)
from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, Optimiser, cases
from crnn_amd import _lib as L
fx=json.load(open(os.path.join(ROOT,"tests","golden","fixtures.json")))
B=65536; rng=np.random.Generator(np.random.PCG64([1234,0])); ts=cases.case2_tsteps(); u0=cases.case2_u0(B,rng)
p=np.array(fx["case2_ckpt"]["p"])
node=NeuralODE(ODEProblem(PRESET_CASE2,ts)); node.set_ensemble(u0,np.abs(rng.standard_normal((B,6,len(ts))))*0.5,np.ones(6))
if os.environ.get("QINDEX"): node.set_queue_order(L.QUEUE_INDEX)
node.train_init(Optimiser(25, PRESET_CASE2), p)
for _ in range(60): node.train_step(want_loss=False)
print(node.train_step(want_loss=True))
