"""case2's experiments, initial parameters and shuffles, re-drawn from the reference's own seeded RNG stream.

TEST INFRASTRUCTURE.  case2/case2.jl seeds Julia's global RNG once (`Random.seed!(1234)`, :11) and then draws, in this order:
  :60      u0_list = rand(Float32, (30, 7))                      -> columns 1:2 scaled to [0.2, 2.2), 3:6 zeroed, 7 -> T in [323, 343) K
  :79      30 x  randn(size(ode_data)) = randn(6, 50)            relative noise 0.05 on the true mechanism's solution (:74-82)
  :86      p = randn(Float32, 25) .* 0.1  (+ 0.8 on ln A and Ea rows, slope 0.1; :87-89)
  :194     randperm(20) at the start of every epoch; :166 randperm(30)[1:1] inside the plot callback every 50th epoch
Nothing else touches the stream (ForwardDiff, OrdinaryDiffEq, Flux and Plots draw nothing), and a restart (`is_restart = true`) re-runs the
first three items before loading the checkpoint, so the data are the same in every session of the run.  With `julia_rng.MersenneTwister`
(Julia 1.6, the version the reference's README names for this case) everything above is reproducible here without Julia.

The true mechanism is solved by the oracle's composite (AutoTsit5(Rosenbrock23), default tolerances abstol 1e-6 / reltol 1e-3 like `solve(prob_trueode,
alg, saveat=tsteps)`), in double; the reference's solve runs on a Float32 state, and the result is rounded to Float32 like `Array(solve(...))` of a
Float32 problem.  What that leaves is visible in the pin: 3e-6 on the recorded loss.
"""
import numpy as np

import julia_rng as J

N_EXP, N_TRAIN, NS, NR, DATASIZE, NOISE = 30, 20, 6, 3, 50, 0.05
PLOT_EVERY = 50


def randperm(rng, n):
    """Julia 1.6 `randperm!(r, a)`: a[1] = 1; for i = 2:n: j = 1 + rand(r, ltm52(i, mask)) [rejection on the masked low bits of a stream
    double]; a[i] = a[j]; a[j] = i; mask grows to 2 mask + 1 when i == 1 + mask.  Returns 1-based experiment numbers like the reference."""
    a = [0] * n
    a[0] = 1
    mask = 3
    for i in range(2, n + 1):
        while True:
            x = rng._next_bits() & mask
            if x <= i - 1:
                break
        j = 1 + x
        if i != j:
            a[i - 1] = a[j - 1]
        a[j - 1] = i
        if i == 1 + mask:
            mask = 2 * mask + 1
    return a


def tsteps():
    """range(0f0, 50f0, length = 50) (case2.jl:64-65): a Float32 range, every element the Float32 nearest to i * 50 / 49."""
    return np.linspace(0.0, 50.0, DATASIZE).astype(np.float32).astype(np.float64)


def draw(orc, cases, n_epochs=100, array_randn=True):
    """Returns dict(u0 [30, 7], ts [50], data [30, 6, 50] (Float32 values), ys [6], p0 [25], perms [n_epochs][20] (1-based), drawn)."""
    rng = J.MersenneTwister(1234, array_randn=array_randn)
    u = rng.rand_f32_array(N_EXP * (NS + 1)).reshape((N_EXP, NS + 1), order="F")
    u0 = np.zeros((N_EXP, NS + 1), dtype=np.float32)                     # Float32 .* Float64 literal -> Float64, stored back into the Float32 array
    u0[:, 0:2] = (u[:, 0:2].astype(np.float64) * 2.0 + 0.2).astype(np.float32)
    u0[:, NS] = (u[:, NS].astype(np.float64) * 20.0 + 323.0).astype(np.float32)
    u0 = u0.astype(np.float64)
    ts = tsteps()
    pbt = orc.make_problem(ns=NS, nr=NR, has_temp=1, lb=1e-300, ub=1e300, inv_R=cases.INV_R, atol=1e-6, rtol=1e-3, solver=2)
    r = orc.solve_batch(pbt, cases.case2_true_theta(), np.ascontiguousarray(u0.T), ts, np.zeros((len(ts), NS, N_EXP)), want_pred=True)
    assert (r["retcode"] == 0).all()
    clean = r["pred"][:, :NS, :].transpose(2, 1, 0).astype(np.float32).astype(np.float64)
    data64 = np.empty_like(clean)
    for i in range(N_EXP):
        z = rng.randn_array(NS * DATASIZE).reshape((NS, DATASIZE), order="F")
        data64[i] = clean[i] + z * clean[i] * NOISE
    ys = cases.max_min(data64, lb=cases.LB_CASE2)                       # from the Float64 ode_data, before the Float32 store (:80-81)
    data = data64.astype(np.float32).astype(np.float64)                  # ode_data_list is Float32
    p0 = rng.randn_f32(NR * (NS + 2) + 1).astype(np.float64) * 0.1
    p0[0:NR] += 0.8
    p0[NR * (NS + 1):NR * (NS + 2)] += 0.8
    p0[-1] = 0.1
    perms = []
    for ep in range(1, n_epochs + 1):
        perms.append(randperm(rng, N_TRAIN))
        if ep % PLOT_EVERY == 0:
            randperm(rng, N_EXP)                                         # the plot callback's draw
    return dict(u0=u0, ts=ts, data=data, ys=ys, p0=p0, perms=perms, drawn=rng.drawn)
