import os, time, json, sys
sys.path.insert(0,'.')
import numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max","/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
from oracle import oracle as orc
from crnn_amd import cases
rng=np.random.Generator(np.random.PCG64(0))
B=16384
u0=cases.case2_u0(B,rng); ts=cases.case2_tsteps()
fx=json.load(open('tests/golden/fixtures.json')); p=np.array(fx['case2_ckpt']['p'])
th,dth=orc.p2vec(2,6,3,p)
pb=orc.make_problem(ns=6,nr=3,has_temp=1,lb=cases.LB_CASE2,ub=10.0,inv_R=cases.INV_R,atol=1e-6,rtol=1e-3,clamp_pred=1)
data=np.zeros((50,6,B))
for nt in (1,8,16,32,64,128,256):
    t0=time.time(); r=orc.solve_batch(pb,th,np.ascontiguousarray(u0.T),ts,data,dtheta=dth,nthreads=nt); dt=time.time()-t0
    print(nt,"threads:",round(B/dt),"traj+grad/s")
