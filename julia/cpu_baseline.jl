# julia/cpu_baseline.jl -- the reference's CPU path, `EnsembleThreads()` over the ensemble of initial conditions, timed on
# the host cores next to the GPU figures (north star; SURVEY 8(d)).
#
# STATUS: WRITTEN, NOT EXECUTED (no Julia in the build image or on the GPU boxes).  It needs the reference's own stack
# (OrdinaryDiffEq, ForwardDiff) and is the harness a Julia-equipped box runs to obtain the true DifferentialEquations.jl
# number; bench.py's `cpu_baseline` is the C restatement ("port") until then.
#
#     julia -t auto julia/cpu_baseline.jl [n_exp]
using OrdinaryDiffEq, ForwardDiff, Random, Statistics
include(joinpath(@__DIR__, "CRNNHip.jl"))
using .CRNNHip

n_exp = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 4096
Random.seed!(1234)
tsteps = collect(range(0.0, 50.0, length = 50))                  # case2/case2.jl:18-19,64-65: 50 points on [0, 50], spacing 50 / 49
prob_dev = CRNNHip.ODEProblem(CRNNHip.PRESET_CASE2, tsteps)      # problem constants (lb, ub, inv_R, tolerances) from the preset
p = randn(25) .* 0.1; p[1:3] .+= 0.8; p[22:24] .+= 0.8; p[25] = 0.1   # case2/case2.jl:85-89
u0_list = zeros(n_exp, 7)
u0_list[:, 1:2] .= rand(n_exp, 2) .* 2.0 .+ 0.2                  # case2/case2.jl:62-65
u0_list[:, 7] .= rand(n_exp) .* 20.0 .+ 323.0

# `ForwardDiff.gradient` pushes Duals through `crnn!`: p2vec of a Dual-valued p = the library's theta(value.(p)) with
# d theta / d p (crnn_p2vec's Jacobian) applied to the partials -- ForwardDiff's own chain rule, clamp / abs conventions
# included (p2vec.hpp follows them).
function CRNNHip.p2vec(prob::CRNNHip.Problem, p::AbstractVector{D}) where {D<:ForwardDiff.Dual}
    th, dth = CRNNHip.p2vec_jac(prob, collect(Float64, ForwardDiff.value.(p)))
    thd = [D(th[k], sum(dth[k, m] * ForwardDiff.partials(p[m]) for m in eachindex(p))) for k in eachindex(th)]
    return CRNNHip.split_theta(prob, thd)
end

rhs!(du, u, p, t) = CRNNHip.crnn!(du, u, p, t, prob_dev)
prob_ref = OrdinaryDiffEq.ODEProblem(rhs!, u0_list[1, :], (tsteps[1], tsteps[end]), p)
ens = EnsembleProblem(prob_ref; prob_func = (pr, i, _) -> remake(pr, u0 = u0_list[i, :]))
alg = AutoTsit5(Rosenbrock23(autodiff = false))                  # case2/case2.jl:26, the script's own (on this model it never leaves Tsit5: a constant
                                                                 # temperature state makes the stiffness estimate NaN; tests/test_case2_stream_pin.py)
alg_stiff = Rosenbrock23(autodiff = false)                       # its stiff branch alone = the stepper the GPU's headline number is quoted on
solve(ens, alg, EnsembleThreads(); trajectories = min(n_exp, 64), saveat = tsteps, abstol = 1e-6, reltol = 1e-3)   # compile
t_solve = @elapsed solve(ens, alg, EnsembleThreads(); trajectories = n_exp, saveat = tsteps, abstol = 1e-6, reltol = 1e-3)

# trajectory + gradient, as the training loop forms it (case2/case2.jl:195): ForwardDiff through the adaptive solver
function loss_of(x, i)
    sol = solve(remake(prob_ref, u0 = u0_list[i, :], p = x), alg; saveat = tsteps, abstol = 1e-6, reltol = 1e-3)
    return mean(abs, clamp.(Array(sol), -10.0, 10.0))            # a data-free stand-in for mae(ode_data, pred)
end
ForwardDiff.gradient(x -> loss_of(x, 1), p)                      # compile
t_grad = @elapsed Threads.@threads for i in 1:n_exp
    ForwardDiff.gradient(x -> loss_of(x, i), p)
end
println("EnsembleThreads() on $(Threads.nthreads()) threads, alg = AutoTsit5(Rosenbrock23): $(n_exp / t_solve) trajectories/s, ",
        "$(n_exp / t_grad) trajectories+gradients/s")
# the same with the stiff branch alone (what bench.py's headline and its `cpu_baseline.forward_tangents_value` correspond to)
function loss_of_stiff(x, i)
    sol = solve(remake(prob_ref, u0 = u0_list[i, :], p = x), alg_stiff; saveat = tsteps, abstol = 1e-6, reltol = 1e-3)
    return mean(abs, clamp.(Array(sol), -10.0, 10.0))
end
ForwardDiff.gradient(x -> loss_of_stiff(x, 1), p)
t_grad_stiff = @elapsed Threads.@threads for i in 1:n_exp
    ForwardDiff.gradient(x -> loss_of_stiff(x, i), p)
end
println("  Rosenbrock23(autodiff = false) alone: $(n_exp / t_grad_stiff) trajectories+gradients/s")
