"""crnn_amd -- MI355X-native (gfx950) implementation of the neural-ODE hot path of
DENG-MIT/CRNN: batched stiff CRNN solve (Rosenbrock23 / Tsit5 / AutoTsit5) + the gradient
ForwardDiff.gradient returns (discrete adjoint or forward tangents) + Flux-style ADAM
update, behind the C ABI in include/crnn_hip.h.

Importing this package loads crnn_amd/csrc/libcrnn_hip.so and fails loudly if it
has not been built (no CPU fallback).
"""
from . import cases, cathode, hychem  # noqa: F401
from ._lib import (JAC_ANALYTIC, JAC_FINITE_DIFF, QUEUE_AUTO, QUEUE_INDEX, GRAD_ADJOINT, GRAD_AUTO, GRAD_FORWARD, LOSS_MAE, LOSS_MSE, PMAP_CASE1, PMAP_CASE2, PMAP_HYCHEM, PMAP_IDENTITY, PMAP_ROBER, PRESET_CASE1, PRESET_HYCHEM,  # noqa: F401
                   PRESET_CASE2, PRESET_ROBER, RET_DTMIN, SOLVER_ROSENBROCK23, SOLVER_TSIT5, SOLVER_AUTOTSIT5, RET_MAXITERS, RET_SUCCESS, RET_UNSTABLE, CrnnError)
from .api import NeuralODE, ODEProblem, Optimiser, crnn, p2vec, p2vec_jac  # noqa: F401
