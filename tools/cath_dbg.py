import json, numpy as np, sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from oracle import oracle as orc
from crnn_amd.cathode import CathodeUQ
from crnn_amd import _lib as L
cfx=json.load(open('tests/golden/fixtures_cathode.json'))
ps=np.array(cfx['theta'])
def two(s):
    dbar, d2bar = np.array(s["dbar"]), np.array(s["d2bar"]); sd = np.sqrt(np.maximum(d2bar - dbar ** 2, 0.0))
    return np.stack([np.array(s["ts"]), dbar + sd, dbar - sd], axis=1)

for (atol,rtol) in ((1e-12,1e-3),(1e-14,1e-9)):
    mk=lambda sv: CathodeUQ([two(s) for s in cfx['sets']],[s['beta'] for s in cfx['sets']],cfx['theta'],atol=atol,rtol=rtol,solver=sv)
    ua, ur = mk(L.SOLVER_AUTOTSIT5), mk(L.SOLVER_ROSENBROCK23)
    rng=np.random.default_rng(12)
    p=1+0.05*rng.standard_normal((40,17)); p[:,6:9]=0; p[0]=1; p[0,6:9]=0
    la,ga,_=ua.solve(p); na=ua.last_stats['n_accept']
    lr,gr,_=ur.solve(p); nr=ur.last_stats['n_accept']
    print('guard',__import__('os').environ.get('CRNN_CATH_GUARD'),'tol',atol,rtol,'steps composite',na,'ros23',nr)
    for n in range(40):
        for i,s in enumerate(cfx['sets']):
            sc=np.max(np.abs(gr[n,i]))
            if np.max(np.abs(ga[n,i]-gr[n,i]))/sc > (3e-2 if rtol>1e-4 else 1e-6): print(n,i,'loss rel %.1e'%(abs(la[n,i]-lr[n,i])/lr[n,i]),'grad vs ros23 %.1e'%(np.max(np.abs(ga[n,i]-gr[n,i]))/sc), ('golden %.1e'%(np.max(np.abs(ga[n,i]-np.array(s['grad'])*ps))/np.max(np.abs(np.array(s['grad'])*ps)))) if n==0 else '')
    ua.close(); ur.close()
