# julia/CRNNHip.jl -- thin `ccall` shim over libcrnn_hip.so (include/crnn_hip.h).
#
# STATUS: WRITTEN, NOT EXECUTED.  Julia is not installed in the build image nor on the GPU
# boxes used for this project (`julia: command not found`), so this file has never been
# run; the identical C ABI is exercised through Python/ctypes (crnn_amd/_lib.py) by the
# test-suite.  It keeps the reference's script-level surface for the hot path
# (case2/case2.jl:91-137,195-197 of DENG-MIT/CRNN):
#
#     p2vec(p) ; crnn!(du,u,p,t) ; prob = ODEProblem(...) ; predict_neuralode(u0, p) ;
#     loss_neuralode(p, i_exp) ; ForwardDiff.gradient(x -> loss_neuralode(x, i_exp), p) ;
#     update!(opt, p, grad)
#
# and adds the batched entry points the reference lacks (SURVEY F3):
#     loss_and_grad(p, idxs) ; train_step!()
module CRNNHip

const LIB = get(ENV, "CRNN_HIP_LIB", joinpath(@__DIR__, "..", "crnn_amd", "csrc", "libcrnn_hip.so"))
const MAX_N = 12

# mirror of `struct crnn_config` (include/crnn_hip.h) -- field order and types must match
mutable struct Config
    abi_version::Int32; ns::Int32; nr::Int32; has_temp::Int32; param_map::Int32; n_save::Int32
    loss_kind::Int32; clamp_pred::Int32; maxiters::Int32; errnorm_sens::Int32; device::Int32; cols_per_lane::Int32
    solver::Int32; grad_mode::Int32; tape_steps::Int32; rhs_kind::Int32
    lb::Float64; ub::Float64; inv_R::Float64; t0::Float64
    atol::NTuple{MAX_N,Float64}; rtol::NTuple{MAX_N,Float64}; rate_scale::NTuple{MAX_N,Float64}
    mw::NTuple{MAX_N,Float64}; gas_const::Float64
    gamma::Float64; qmin::Float64; qmax::Float64; beta1::Float64; beta2::Float64
    qsteady_min::Float64; qsteady_max::Float64; qoldinit::Float64; dtmin::Float64
    Config() = new()
end

mutable struct Stats
    n_traj::Int64; n_ok::Int64; n_accept::Int64; n_reject::Int64; kernel_ms::Float64
    Stats() = new(0, 0, 0, 0, 0.0)
end

mutable struct OptConfig
    use_expdecay::Int32; decay_step::Int32; ed_eta0::Float64; ed_decay::Float64; ed_clip::Float64
    eta::Float64; beta1::Float64; beta2::Float64; wd::Float64; grad_clip_norm::Float64
    OptConfig() = new()
end

const PRESET_CASE1, PRESET_CASE2, PRESET_ROBER, PRESET_HYCHEM = Int32(1), Int32(2), Int32(3), Int32(4)
const GRAD_AUTO, GRAD_FORWARD, GRAD_ADJOINT = Int32(0), Int32(1), Int32(2)   # cfg.grad_mode: how ForwardDiff.gradient is formed
# alg = Rosenbrock23() (rober_crnn.jl:33) / Tsit5() (case1.jl:28) / AutoTsit5(Rosenbrock23()) (case2.jl:26)
const ROSENBROCK23, TSIT5, AUTOTSIT5 = Int32(0), Int32(1), Int32(2)

check(rc, ctx=C_NULL) = rc == 0 ? nothing :
    error(unsafe_string(ccall((:crnn_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx)))

"""`prob = ODEProblem(crnn, u0, tspan, saveat=tsteps, atol=atol, rtol=rtol)` (case2/case2.jl:120-121)."""
mutable struct Problem
    cfg::Config
    ctx::Ptr{Cvoid}
    tsteps::Vector{Float64}
    B::Int
    adhoc::Union{Nothing,Problem}      # one-IC context behind predict_neuralode(prob, u0, p)
end

function ODEProblem(preset::Integer, tsteps::AbstractVector; atol=nothing, rtol=nothing, rate_scale=nothing, device=0, alg=nothing,
                    cfg::Union{Nothing,Config}=nothing, errnorm_sens=nothing)
    if cfg === nothing
        cfg = Config()
        check(ccall((:crnn_config_preset, LIB), Int32, (Ref{Config}, Int32), cfg, preset))
    else
        cfg = deepcopy(cfg)            # a second context with the same problem constants (predict_neuralode's one-IC context)
    end
    # errnorm_sens = 1: the step-size controller sees ForwardDiff's dual-inclusive error norm, i.e. a gradient call
    # takes the step sequence `ForwardDiff.gradient` through the adaptive solver takes; squared norm / length(u) (early DiffEqBase 6).
    # errnorm_sens = 2: / totallength(u) -- the form the reference's case2 checkpoint history was recorded with (tests/test_case2_stream_pin.py)
    # and the cathode Manifest pins: the reference-faithful mode.
    errnorm_sens === nothing || (cfg.errnorm_sens = errnorm_sens)
    alg === nothing || check(ccall((:crnn_config_set_solver, LIB), Int32, (Ref{Config}, Int32), cfg, alg))   # also sets the PI exponents
    cfg.n_save = length(tsteps); cfg.device = device
    n = cfg.ns + cfg.has_temp
    atol === nothing || (cfg.atol = ntuple(i -> i <= n ? Float64(atol isa Number ? atol : atol[i]) : 0.0, MAX_N))
    rtol === nothing || (cfg.rtol = ntuple(i -> i <= n ? Float64(rtol isa Number ? rtol : rtol[i]) : 0.0, MAX_N))
    rate_scale === nothing || (cfg.rate_scale = ntuple(i -> i <= cfg.ns ? Float64(rate_scale[i]) : 1.0, MAX_N))
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:crnn_ctx_create, LIB), Int32, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, ctx))
    prob = Problem(cfg, ctx[], collect(Float64, tsteps), 0, nothing)
    finalizer(p -> ccall((:crnn_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), p.ctx), prob)
    return prob
end

"""Upload `u0_list[n_exp, n]`, `ode_data_list[n_exp, n_obs, datasize]`, `yscale` (case2/case2.jl:62-83).
Julia's column-major layout IS the ABI's IC-fastest layout: no copy, no transpose."""
function set_ensemble!(prob::Problem, u0_list::Matrix{Float64}, ode_data_list::Array{Float64,3}, yscale::Vector{Float64};
                       i_obs::Union{Nothing,Vector{Int32}}=nothing)
    B = size(u0_list, 1)
    n_obs = i_obs === nothing ? prob.cfg.ns : length(i_obs)
    GC.@preserve u0_list ode_data_list yscale i_obs begin
        check(ccall((:crnn_ctx_set_data, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Int32, Int64),
                    prob.ctx, u0_list, ode_data_list, prob.tsteps, yscale,
                    i_obs === nothing ? C_NULL : pointer(i_obs), n_obs, B), prob.ctx)
    end
    prob.B = B
end

"""HyChem (HyChem/crnn_pyrolysis_mass.jl:44-47,103-104): `Tlist[n_exp, D]`, `Plist[n_exp, D]` on `tsteps`, piecewise linear in t."""
set_tables!(prob::Problem, Tlist::Matrix{Float64}, Plist::Matrix{Float64}) =
    check(ccall((:crnn_ctx_set_tables, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), prob.ctx, Tlist, Plist), prob.ctx)

"""Queue order of the adjoint kernels: `QUEUE_AUTO = 0` (by the previous launch's step counts, the default) or
`QUEUE_INDEX = 1` (index order: batch sums depend on the call's inputs alone, bit for bit).  include/crnn_hip.h."""
const QUEUE_AUTO = Int32(0); const QUEUE_INDEX = Int32(1)
set_queue_order!(prob::Problem, order::Integer) =
    check(ccall((:crnn_ctx_set_queue_order, LIB), Int32, (Ptr{Cvoid}, Int32), prob.ctx, Int32(order)), prob.ctx)
"""Lanes per trajectory in the Rosenbrock23 adjoint kernel: 0 = auto (default), 1, 2 (a lane pair per trajectory: shards
smaller than the chip, e.g. one GPU's share of a strongly-scaled batch)."""
last_lanes_per_traj(prob::Problem) = ccall((:crnn_last_lanes_per_traj, LIB), Int32, (Ptr{Cvoid},), prob.ctx)
tape_retries(prob::Problem) = ccall((:crnn_tape_retries, LIB), Int64, (Ptr{Cvoid},), prob.ctx)
hychem_block_cap(prob::Problem) = ccall((:crnn_hychem_block_cap, LIB), Int32, (Ptr{Cvoid},), prob.ctx)
set_lanes_per_traj!(prob::Problem, lanes::Integer) =
    check(ccall((:crnn_ctx_set_lanes_per_traj, LIB), Int32, (Ptr{Cvoid}, Int32), prob.ctx, Int32(lanes)), prob.ctx)

"""`set_jacobian!(prob, JAC_FINITE_DIFF)`: the primal launches build `W` from forward differences of the right-hand side, as
`Rosenbrock23(autodiff=false)` does (`case2/case2.jl:26`); on a HyChem problem (`HyChem/crnn_pyrolysis_mass.jl:29`) also the
finite-difference time derivative on the `T(t)`, `P(t)` tables, for Rosenbrock23 and inside `AutoTsit5(Rosenbrock23)`;
`JAC_ANALYTIC` (default): the exact Jacobian (`autodiff=true`)."""
set_jacobian!(prob::Problem, mode::Integer) =
    check(ccall((:crnn_ctx_set_jacobian, LIB), Int32, (Ptr{Cvoid}, Int32), prob.ctx, Int32(mode)), prob.ctx)
const JAC_ANALYTIC = Int32(0)
const JAC_FINITE_DIFF = Int32(1)

"""`sol.destats.naccept / nreject` of every `solve` of the most recent ensemble launch over `first .+ (0:count-1)` (0-based)."""
function last_step_counts(prob::Problem, first::Integer = 0, count::Integer = prob.B - first)
    na = zeros(Int32, count); nr = zeros(Int32, count)
    check(ccall((:crnn_last_step_counts, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Int32}, Ptr{Int32}),
                prob.ctx, Int64(first), Int64(count), na, nr), prob.ctx)
    return na, nr
end

"""`(violations, first_site) = debug_bounds()`: diagnostic builds of the library (-DCRNN_BOUNDS_CHECK) count out-of-range
accesses of the adjoint kernels; `nothing` on a release build."""
function debug_bounds()
    v = Ref{UInt32}(0); s = Ref{UInt32}(0)
    rc = ccall((:crnn_debug_bounds, LIB), Int32, (Ref{UInt32}, Ref{UInt32}), v, s)
    return rc == 0 ? (v[], s[]) : nothing
end

"""`w_in, w_b, w_out = p2vec(p)` (case2/case2.jl:91-99) plus the Jacobian d theta / d p."""
function p2vec_jac(prob::Problem, p::Vector{Float64})
    c = prob.cfg
    nth = ccall((:crnn_config_n_theta, LIB), Int32, (Ref{Config},), c)
    th = zeros(nth); dth = zeros(nth, length(p))
    check(ccall((:crnn_p2vec, LIB), Int32, (Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                c.param_map, c.ns, c.nr, p, th, dth))
    return th, dth
end
"""theta = [vec(w_in); w_b; vec(w_out)] (column-major, as the library packs it) -> the three arrays `p2vec` returns."""
function split_theta(prob::Problem, th::AbstractVector)
    c = prob.cfg; n = c.ns + (c.rhs_kind == 1 ? 2 : c.has_temp)   # feature rows of w_in
    return reshape(th[1:n*c.nr], n, c.nr), th[n*c.nr+1:(n+1)*c.nr], reshape(th[(n+1)*c.nr+1:end], c.ns, c.nr)
end
p2vec(prob::Problem, p::AbstractVector{<:AbstractFloat}) = split_theta(prob, p2vec_jac(prob, collect(Float64, p))[1])
# A number type that carries derivatives (ForwardDiff.Dual when `ForwardDiff.gradient` differentiates through a CPU solve of
# `crnn!`) needs its own method: theta(value.(p)) from the library and d theta / d p (p2vec_jac) pushed through the
# partials.  julia/cpu_baseline.jl adds it for ForwardDiff.Dual; this module itself does not depend on ForwardDiff.

function _solve(prob::Problem, th, dth, first, count, sample; want_pred=false)
    n = prob.cfg.ns + prob.cfg.has_temp; D = length(prob.tsteps); B = prob.B
    ndir = dth === nothing ? 0 : size(dth, 2)
    pred = want_pred ? zeros(B, n, D) : Float64[]
    loss = zeros(B); grad = zeros(max(ndir, 1)); ret = zeros(Int32, B); nsv = zeros(Int32, B); st = Stats()
    GC.@preserve th dth pred loss grad ret nsv begin
        check(ccall((:crnn_solve, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Int64, Int64, Int32, Ptr{Float64}, Ptr{Float64},
                     Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ref{Stats}),
                    prob.ctx, th, dth === nothing ? C_NULL : pointer(dth), ndir, first, count, sample,
                    want_pred ? pointer(pred) : C_NULL, loss, ndir > 0 ? pointer(grad) : C_NULL, ret, nsv, st), prob.ctx)
    end
    any(ret[first+1:first+count] .!= 0) && println("ode solver failed")   # robertson/rober_crnn.jl:130-134
    return pred, loss, grad[1:ndir], ret, nsv, st
end

"""`crnn!(du, u, p, t)`: the CPU definition of the right-hand side (case2/case2.jl:114-118, case1/case1.jl:80-83,
robertson/rober_crnn.jl:113-116) with the weights `p2vec` returns -- what a Julia host hands to DifferentialEquations.jl's
own `ODEProblem` (the CPU reference run below); the device never calls it.  `prob` supplies lb, ub, inv_R, dydt_scale."""
function crnn!(du::AbstractVector, u::AbstractVector, p::AbstractVector, t, prob::Problem)
    c = prob.cfg; ns = Int(c.ns); nr = Int(c.nr)
    w_in, w_b, w_out = p2vec(prob, p)                 # eltype(p) may be a Dual (p2vec above)
    z = Vector{promote_type(eltype(w_b), eltype(u))}(w_b)
    for j in 1:nr
        for i in 1:ns
            z[j] += w_in[i, j] * log(clamp(u[i], c.lb, c.ub))
        end
        c.has_temp == 1 && (z[j] += w_in[ns+1, j] * (c.inv_R / u[ns+1]))
    end
    r = exp.(z)
    for i in 1:ns
        du[i] = c.rate_scale[i] * sum(w_out[i, j] * r[j] for j in 1:nr)
    end
    c.has_temp == 1 && (du[ns+1] = zero(eltype(du)))
    return du
end

"""`pred = predict_neuralode(u0, p)` (case2/case2.jl:124-128): `clamp.(Array(solve(prob, alg, u0=u0, p=p)), -ub, ub)`,
[n, D]; columns beyond a failed solve's last saved point are zero and "ode solver failed" is printed
(robertson/rober_crnn.jl:130-134).  Integrated on the device through a one-IC context created on first use."""
function predict_neuralode(prob::Problem, u0::AbstractVector, p::AbstractVector; sample=length(prob.tsteps))
    if prob.adhoc === nothing
        c = prob.cfg
        a = ODEProblem(Int32(0), prob.tsteps; cfg=c)
        prob.adhoc = a
    end
    a = prob.adhoc
    n = a.cfg.ns + a.cfg.has_temp
    set_ensemble!(a, reshape(collect(Float64, u0), 1, n), zeros(1, Int(a.cfg.ns), length(a.tsteps)), ones(Int(a.cfg.ns)))
    th, _ = p2vec_jac(a, collect(Float64, p))
    pred, _, _, _, nsv, _ = _solve(a, th, nothing, 0, 1, sample; want_pred=true)
    return reshape(pred, n, length(a.tsteps))[:, 1:nsv[1]]      # B = 1: [1, n, D] column-major = [n, D]
end

"""`loss_neuralode(p, i_exp)` (case2/case2.jl:132-137); `i_exp` is 1-based like the reference.  The reference's function
reads `u0_list`, `ode_data_list`, `yscale` from globals; here they live behind `prob` (set_ensemble!), hence the extra
leading argument -- the only change to the reference's signature (INTEGRATION.md)."""
function loss_neuralode(prob::Problem, p, i_exp; sample=length(prob.tsteps))
    th, _ = p2vec_jac(prob, p)
    _, loss, = _solve(prob, th, nothing, i_exp - 1, 1, sample)
    return loss[i_exp]
end

"""`ForwardDiff.gradient(x -> loss_neuralode(x, i_exp), p)` (case2/case2.jl:195)."""
function gradient(prob::Problem, p, i_exp; sample=length(prob.tsteps))
    th, dth = p2vec_jac(prob, p)
    _, _, grad, = _solve(prob, th, dth, i_exp - 1, 1, sample)
    return grad
end

"""Mean loss over experiments `idxs` (a contiguous range) and its gradient, one launch, device-side p2vec."""
function loss_and_grad(prob::Problem, p::Vector{Float64}, idxs::UnitRange=1:prob.B; sample=length(prob.tsteps))
    loss = Ref{Float64}(0.0); grad = zeros(length(p)); st = Stats()
    check(ccall((:crnn_loss_grad, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int32, Ref{Float64}, Ptr{Float64}, Ref{Stats}),
                prob.ctx, p, first(idxs) - 1, length(idxs), sample, loss, grad, st), prob.ctx)
    return loss[], grad
end

"""`opt = Flux.Optimiser(ExpDecay(...), ADAMW(...))` + `update!(opt, p, grad)` (case2/case2.jl:31-32,197)."""
mutable struct Optimiser
    cfg::OptConfig
    state::Vector{Float64}
end
function Optimiser(preset::Integer, np::Integer)
    o = OptConfig()
    check(ccall((:crnn_opt_preset, LIB), Int32, (Ref{OptConfig}, Int32), o, preset))
    st = zeros(ccall((:crnn_opt_state_len, LIB), Int32, (Int32,), np))
    check(ccall((:crnn_opt_init, LIB), Int32, (Ref{OptConfig}, Int32, Ptr{Float64}), o, np, st))
    return Optimiser(o, st)
end
update!(opt::Optimiser, p::Vector{Float64}, grad::Vector{Float64}) =
    check(ccall((:crnn_opt_update, LIB), Int32, (Ref{OptConfig}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                opt.cfg, length(p), p, grad, opt.state))

"""`update!(opt, p, grad)` on the device-resident `p` with a caller-supplied gradient (case2/case2.jl:197)."""
train_update!(prob::Problem, grad::Vector{Float64}) =
    check(ccall((:crnn_train_update, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), prob.ctx, grad), prob.ctx)
function params(prob::Problem, np::Integer)
    p = zeros(np)
    check(ccall((:crnn_get_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), prob.ctx, p), prob.ctx)
    return p
end
"""The device-resident optimiser state -- what `@save "./checkpoint/mymodel.bson" p opt ...` / `@load` keep across a
restart (case2/case2.jl:178-187,213): [m | v | beta1^t, beta2^t, eta_expdecay, ncalls]."""
function opt_state(prob::Problem, np::Integer)
    st = zeros(ccall((:crnn_opt_state_len, LIB), Int32, (Int32,), np))
    check(ccall((:crnn_get_opt_state, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), prob.ctx, st), prob.ctx)
    return st
end
set_opt_state!(prob::Problem, st::Vector{Float64}) =
    check(ccall((:crnn_set_opt_state, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), prob.ctx, st), prob.ctx)

"""Multi-GPU, one Julia process per GPU: either the library's own RCCL communicator (`comm_init!` with the unique id
rank 0 obtained from `comm_unique_id()` and broadcast by the host's launcher, e.g. MPI.bcast), or the host's collective
handed in as a C callback (`set_allreduce!(prob, @cfunction(my_allreduce, Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid})))`)."""
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:crnn_comm_get_unique_id, LIB), Int32, (Ptr{UInt8},), id))
    return id
end
comm_init!(prob::Problem, id::Vector{UInt8}, rank::Integer, world::Integer) =
    check(ccall((:crnn_comm_init, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), prob.ctx, id, rank, world), prob.ctx)
set_allreduce!(prob::Problem, fn::Ptr{Cvoid}, user::Ptr{Cvoid}=C_NULL) =
    check(ccall((:crnn_comm_set_allreduce, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), prob.ctx, fn, user), prob.ctx)

"""Device-resident training (p and optimiser state stay in HBM; one all-reduce per step when a communicator is attached)."""
train_init!(prob::Problem, opt::Optimiser, p0::Vector{Float64}) =
    check(ccall((:crnn_train_init, LIB), Int32, (Ptr{Cvoid}, Ref{OptConfig}, Ptr{Float64}), prob.ctx, opt.cfg, p0), prob.ctx)
function train_step!(prob::Problem; idxs::UnitRange=1:prob.B, sample=length(prob.tsteps))
    loss = Ref{Float64}(0.0)
    check(ccall((:crnn_train_step, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int32, Ref{Float64}),
                prob.ctx, first(idxs) - 1, length(idxs), sample, loss), prob.ctx)
    return loss[]
end

# ---------------------------------------------------------------------------------------------------------
# Bayesian cathode ensemble (Cathode_NCM333_UQ/src_333/network.jl:196-275): all particles x heating rates in one launch
# ---------------------------------------------------------------------------------------------------------
mutable struct CathodeConfig
    abi_version::Int32; device::Int32; maxiters::Int32; grad_mode::Int32
    lb_clamp::Float64; T0::Float64; atol::Float64; rtol::Float64
    gamma::Float64; qmin::Float64; qmax::Float64; beta1::Float64; beta2::Float64
    qsteady_min::Float64; qsteady_max::Float64; qoldinit::Float64
    CathodeConfig() = new()
end

mutable struct Cathode
    ctx::Ptr{Cvoid}; n_sets::Int; Dmax::Int; p_scales::Vector{Float64}
end

"""`l_exp_data[i]` = [t, replicas...] per heating rate (dataset.jl:19-23), `heating_rates` in K/min, `p_scales` = the optimum the particles are scaled by."""
function Cathode(l_exp_data::Vector{Matrix{Float64}}, heating_rates::Vector{Float64}, p_scales::Vector{Float64}; errnorm_sens::Integer=2)
    cfg = CathodeConfig()
    check(ccall((:crnn_cathode_config_default, LIB), Int32, (Ref{CathodeConfig},), cfg))
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:crnn_cathode_create, LIB), Int32, (Ref{CathodeConfig}, Ref{Ptr{Cvoid}}), cfg, ctx))
    ns = length(l_exp_data); D = Int32[size(e, 1) for e in l_exp_data]; Dmax = maximum(D)
    ts = zeros(Dmax, ns); dbar = zeros(Dmax, ns); d2bar = zeros(Dmax, ns)      # column-major [Dmax, ns] = row-major [ns][Dmax]
    for (s, e) in enumerate(l_exp_data)
        n = size(e, 1)
        ts[1:n, s] = e[:, 1]; ts[n+1:end, s] = e[end, 1] .+ (1:Dmax-n)
        dbar[1:n, s] = vec(sum(e[:, 2:end], dims=2)) ./ (size(e, 2) - 1)
        d2bar[1:n, s] = vec(sum(e[:, 2:end] .^ 2, dims=2)) ./ (size(e, 2) - 1)
    end
    rc = ccall((:crnn_cathode_set_obs, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
               ctx[], ns, Dmax, D, ts, dbar, d2bar, heating_rates)
    rc == 0 || error(unsafe_string(ccall((:crnn_cathode_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx[])))
    c = Cathode(ctx[], ns, Dmax, p_scales[1:17])
    finalizer(x -> ccall((:crnn_cathode_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.ctx), c)
    # errnorm_sens = 2 (default): gradient calls as ForwardDiff.gradient evaluates them (network.jl:232) -- chunks of 9 + 8 partials, every
    # chunk its own adaptive solve with the partials in the error norm; the robust gradient of this model.  0: the primal-norm adjoint
    # (2.2x faster; off by orders of magnitude on a few per cent of the trajectories of a particle cloud: profiles/r04m)
    errnorm_sens == 0 || set_errnorm_sens!(c, errnorm_sens)
    return c
end

"""The loop body of `dlnprob(p, i_exp)` (network.jl:227-252) for ALL particles and heating rates: loss[n_sets, N], grad_p[17, n_sets, N]."""
function solve(c::Cathode, p::Matrix{Float64})          # p [N, 17] normalised particles
    N = size(p, 1)
    theta = permutedims(p[:, 1:17] .* c.p_scales')        # [17, N] column-major = row-major [N][17]
    loss = zeros(c.n_sets, N); grad = zeros(17, c.n_sets, N); ret = zeros(Int32, c.n_sets, N); nsv = zeros(Int32, c.n_sets, N); st = Stats()
    rc = ccall((:crnn_cathode_solve, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ref{Stats}),
               c.ctx, theta, N, loss, grad, C_NULL, ret, nsv, st)
    rc == 0 || error(unsafe_string(ccall((:crnn_cathode_last_error, LIB), Cstring, (Ptr{Cvoid},), c.ctx)))
    any(ret .!= 0) && println("ode solver failed")
    return loss, grad .* c.p_scales                       # d loss / d p through p .* p_scales
end

"""The SVGD move (network.jl:67-87 `svgd_kernel` + crnn_cathode.jl:36-50) on the device: returns p_new [N, dim] and the bandwidth h."""
function svgd_update(p::Matrix{Float64}, lnpgrad::Matrix{Float64}, stepsize::Float64; h::Float64=-1.0, device::Integer=0)
    N, dim = size(p)
    pr = permutedims(p); gr = permutedims(lnpgrad)       # row-major [N][dim] for the ABI
    pn = similar(pr); hout = Ref(0.0)
    check(ccall((:crnn_svgd_update, LIB), Int32, (Int32, Ptr{Float64}, Ptr{Float64}, Int64, Int32, Float64, Float64, Ptr{Float64}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}),
                device, pr, gr, N, dim, stepsize, h, pn, hout, C_NULL, C_NULL))
    return permutedims(pn), hout[]
end

"""Adjoint tape layout: 1 = every step in full (default, fastest), 4 / 8 = checkpointed (2.5x / 3.3x less tape traffic and memory)."""
set_tape_every!(c::Cathode, every::Integer) = ccall((:crnn_cathode_set_tape_every, LIB), Int32, (Ptr{Cvoid}, Int32), c.ctx, Int32(every)) == 0 ||
    error(unsafe_string(ccall((:crnn_cathode_last_error, LIB), Cstring, (Ptr{Cvoid},), c.ctx)))

"""Stepper of the primal calls (`solve(c, p)` without gradients is not exposed here; `pred_n_ode` / `loss_neuralode` go through it):
0 = Rosenbrock23 (default), 1 = `AutoTsit5(TRBDF2(autodiff = true))` -- the reference's `alg` (network.jl:195) --, 2 = AutoTsit5 with
Rosenbrock23 as the stiff algorithm.  Gradient launches always run the Rosenbrock23 adjoint."""
set_solver!(c::Cathode, solver::Integer) = ccall((:crnn_cathode_set_solver, LIB), Int32, (Ptr{Cvoid}, Int32), c.ctx, Int32(solver)) == 0 ||
    error(unsafe_string(ccall((:crnn_cathode_last_error, LIB), Cstring, (Ptr{Cvoid},), c.ctx)))

"""The gradient as the reference evaluates it (network.jl:232: `ForwardDiff.gradient` through the adaptive solve): `mode` 1 / 2 puts the
partials of ForwardDiff's chunks (9, then 8 + a zero partial) into the error norm of gradient launches (2 = the `totallength(u)` divisor of
the DiffEqBase this project pins); `mode` 0 (default): primal-only norm, discrete adjoint."""
set_errnorm_sens!(c::Cathode, mode::Integer) = ccall((:crnn_cathode_set_errnorm_sens, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}), c.ctx, Int32(mode), c.p_scales) == 0 ||
    error(unsafe_string(ccall((:crnn_cathode_last_error, LIB), Cstring, (Ptr{Cvoid},), c.ctx)))

"""Device-resident SVGD loop (crnn_cathode.jl:36-50): `set_particles!(c, p)` uploads the normalised particles p [N, 17];
`svgd_step!(c, i_exp, normalizer2, stepsize)` runs dlnprob for heating rate i_exp and the SVGD move on the device
(returns the mean loss and the bandwidth); `particles(c, N)` copies them out."""
function set_particles!(c::Cathode, p::Matrix{Float64})
    pr = permutedims(p)                                   # row-major [N][17] for the ABI
    rc = ccall((:crnn_cathode_set_particles, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), c.ctx, pr, c.p_scales, size(p, 1))
    rc == 0 || error(unsafe_string(ccall((:crnn_cathode_last_error, LIB), Cstring, (Ptr{Cvoid},), c.ctx)))
    return size(p, 1)
end
function svgd_step!(c::Cathode, i_exp::Integer, normalizer2::Vector{Float64}, stepsize::Float64; h::Float64=-1.0)   # normalizer2[k] = Normalizer[i_exp, col(k)]^2, 17 entries
    loss = Ref(0.0); hout = Ref(0.0)
    rc = ccall((:crnn_cathode_svgd_step, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Float64, Float64, Ref{Float64}, Ref{Float64}, Ptr{Float64}),
               c.ctx, Int32(i_exp - 1), normalizer2, stepsize, h, loss, hout, C_NULL)
    rc == 0 || error(unsafe_string(ccall((:crnn_cathode_last_error, LIB), Cstring, (Ptr{Cvoid},), c.ctx)))
    return loss[], hout[]
end
function particles(c::Cathode, N::Integer)
    pr = zeros(17, N)
    rc = ccall((:crnn_cathode_get_particles, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), c.ctx, pr)
    rc == 0 || error(unsafe_string(ccall((:crnn_cathode_last_error, LIB), Cstring, (Ptr{Cvoid},), c.ctx)))
    return permutedims(pr)
end

# ---------------------------------------------------------------------------------------------------------------------------
# The rest of include/crnn_hip.h: load-time checks of this file's struct mirrors, queries, the two-phase training step for hosts
# that own their collective (MPI.jl), RCCL teardown, the cathode's particle exchange.  Same status as everything above: not executed.

"""`check_abi()`: refuses a library whose ABI version or struct sizes differ from this file's mirrors (`crnn_abi_version`, `crnn_sizeof`
for `crnn_config` / `crnn_stats` / `crnn_opt_config` / `crnn_cathode_config`).  Call once after loading; `build_info()` names the sources
the binary was compiled from (`"src=<16 hex digits> arch=gfx950"`)."""
function check_abi()
    v = ccall((:crnn_abi_version, LIB), Int32, ())
    v == 5 || error("libcrnn_hip.so has ABI version $v, CRNNHip.jl is written against 5")
    for (which, T) in ((0, Config), (1, Stats), (2, OptConfig), (3, CathodeConfig))
        n = ccall((:crnn_sizeof, LIB), Int32, (Int32,), Int32(which))
        n == sizeof(T) || error("sizeof mismatch for $T: library $n, Julia mirror $(sizeof(T))")
    end
    return true
end
build_info() = unsafe_string(ccall((:crnn_build_info, LIB), Cstring, ()))

"""Lengths of `p` and of the effective weights `theta` for a parameter map (`p2vec`'s input and output, case2/case2.jl:91-99)."""
n_params(param_map::Integer, ns::Integer, nr::Integer) = Int(ccall((:crnn_n_params, LIB), Int32, (Int32, Int32, Int32), param_map, ns, nr))
n_theta(ns::Integer, nr::Integer, extra_rows::Integer) = Int(ccall((:crnn_n_theta, LIB), Int32, (Int32, Int32, Int32), ns, nr, extra_rows))

"""`sol.destats` of the most recent ensemble launch, summed over its trajectories (synchronises)."""
function last_stats(prob::Problem)
    st = Stats()
    check(ccall((:crnn_last_stats, LIB), Int32, (Ptr{Cvoid}, Ref{Stats}), prob.ctx, st), prob.ctx)
    return st
end
"""HIP-event durations (ms) of the solve kernel of the last `n <= 64` launches, oldest first (synchronises)."""
function kernel_times(prob::Problem, n::Integer)
    ms = zeros(n)
    check(ccall((:crnn_kernel_times, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int32), prob.ctx, ms, Int32(n)), prob.ctx)
    return ms
end
synchronize(prob::Problem) = check(ccall((:crnn_synchronize, LIB), Int32, (Ptr{Cvoid},), prob.ctx), prob.ctx)
"""Run all of the context's work on a caller-owned `hipStream_t` (e.g. AMDGPU.jl's stream handle as a `Ptr{Cvoid}`)."""
set_stream!(prob::Problem, hip_stream::Ptr{Cvoid}) = check(ccall((:crnn_ctx_set_stream, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), prob.ctx, hip_stream), prob.ctx)

"""`set_ensemble_device!(prob, d_u0, d_data, yscale, B)`: as `set_ensemble!`, from buffers already resident on the context's device
(`d_u0` [n][B], `d_data` [D][n_obs][B], both IC-fastest `Float64`; e.g. `pointer(::ROCArray)`): no PCIe copy."""
function set_ensemble_device!(prob::Problem, d_u0::Ptr{Cvoid}, d_data::Ptr{Cvoid}, yscale::Vector{Float64}, B::Integer;
                              i_obs::Union{Nothing,Vector{Int32}}=nothing)
    nobs = i_obs === nothing ? Int32(prob.cfg.ns) : Int32(length(i_obs))
    check(ccall((:crnn_ctx_set_data_device, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Int32, Int64),
                prob.ctx, d_u0, d_data, prob.tsteps, yscale, i_obs === nothing ? C_NULL : i_obs, nobs, Int64(B)), prob.ctx)
    prob.B = B
    return prob
end

"""Overwrite the device-resident parameters of the training loop (`p .= p_new` between `train_step!`s; the optimiser state stays)."""
set_params!(prob::Problem, p::Vector{Float64}) = check(ccall((:crnn_set_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), prob.ctx, p), prob.ctx)

"""The two halves of `train_step!` for a host that runs its own collective between them (MPI.jl: `Allreduce!` on the device vector
`grad_buffer(prob)` = `[grad_sum(P) | n_overflow | loss_sum, n_ok, n_accept, n_reject, n_traj]`, P + 6 doubles):
`train_step_begin!(prob, first, count)` launches solve + gradient + reduction, `train_step_end!(prob)` applies `update!` and returns
the mean loss over all ranks' trajectories."""
train_step_begin!(prob::Problem, first::Integer=0, count::Integer=prob.B; n_save_active::Integer=length(prob.tsteps)) =
    check(ccall((:crnn_train_step_begin, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int32), prob.ctx, Int64(first), Int64(count), Int32(n_save_active)), prob.ctx)
function train_step_end!(prob::Problem)
    loss = Ref(0.0)
    check(ccall((:crnn_train_step_end, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}), prob.ctx, loss), prob.ctx)
    return loss[]
end
function grad_buffer(prob::Problem)
    dptr = Ref{Ptr{Cvoid}}(C_NULL); n = Ref{Int32}(0)
    check(ccall((:crnn_grad_buffer, LIB), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Int32}), prob.ctx, dptr, n), prob.ctx)
    return dptr[], Int(n[])
end

"""RCCL side of a multi-GPU run (one process per GPU): `comm_destroy!` tears the communicator of `comm_init!` down; `comm_collectives`
is the number of all-reduces the training loop has issued (every rank must report the same number); `allreduce_grad!(prob, buf)` sums
a host vector in place over the ranks."""
comm_destroy!(prob::Problem) = check(ccall((:crnn_comm_destroy, LIB), Int32, (Ptr{Cvoid},), prob.ctx), prob.ctx)
comm_collectives(prob::Problem) = ccall((:crnn_comm_collectives, LIB), Int64, (Ptr{Cvoid},), prob.ctx)
allreduce_grad!(prob::Problem, buf::Vector{Float64}) =
    (check(ccall((:crnn_allreduce_grad, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int32), prob.ctx, buf, Int32(length(buf))), prob.ctx); buf)

"""Cathode ensemble over several GPUs (crnn_cathode.jl:31: every rank needs all particles' `lnpgrad` for the SVGD move): the N particles are
partitioned contiguously over the ranks; `allgather(c, local, n_total)` passes this rank's rows `local` [n_local, width] and returns all
`n_total` rows (one `ncclAllGather`; without a communicator `n_local == n_total` and the rows are copied)."""
function cathode_check(c::Cathode, rc)
    rc == 0 || error(unsafe_string(ccall((:crnn_cathode_last_error, LIB), Cstring, (Ptr{Cvoid},), c.ctx)))
    return nothing
end
comm_init!(c::Cathode, id::Vector{UInt8}, rank::Integer, world::Integer) =
    cathode_check(c, ccall((:crnn_cathode_comm_init, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), c.ctx, id, Int32(rank), Int32(world)))
comm_destroy!(c::Cathode) = cathode_check(c, ccall((:crnn_cathode_comm_destroy, LIB), Int32, (Ptr{Cvoid},), c.ctx))
function allgather(c::Cathode, local_rows::Matrix{Float64}, n_total::Integer)
    n_local, width = size(local_rows)
    lr = permutedims(local_rows)                          # row-major [n_local][width] for the ABI
    full = zeros(width, n_total)
    cathode_check(c, ccall((:crnn_cathode_allgather, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int32, Int64, Ptr{Float64}),
                           c.ctx, lr, Int64(n_local), Int32(width), Int64(n_total), full))
    return permutedims(full)
end
"""`(accepted_1, rejected_1, accepted_2, rejected_2)` of the last `errnorm_sens` gradient call: ForwardDiff's two chunks (9 + 8) are
separate adaptive solves with their own step counts."""
function last_chunk_stats(c::Cathode)
    out = zeros(Int64, 4)
    cathode_check(c, ccall((:crnn_cathode_last_chunk_stats, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), c.ctx, out))
    return Tuple(out)
end

end # module
