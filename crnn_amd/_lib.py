"""ctypes binding of the C ABI in include/crnn_hip.h (libcrnn_hip.so, gfx950).

The product path has no CPU fallback: if the HIP library is missing this module
raises at import time, and every compute entry point fails loudly when no
MI355X is visible (crnn_ctx_create returns an error).
"""
from __future__ import annotations

import ctypes as C
import os

MAX_N = 12
MAX_NR = 16
ABI_VERSION = 5
UNIQUE_ID_BYTES = 128

PMAP_IDENTITY, PMAP_CASE1, PMAP_CASE2, PMAP_ROBER, PMAP_HYCHEM = 0, 1, 2, 3, 4
RHS_CRNN, RHS_HYCHEM = 0, 1
LOSS_MAE, LOSS_MSE = 0, 1
RET_SUCCESS, RET_MAXITERS, RET_DTMIN, RET_UNSTABLE = 0, 1, 2, 3
PRESET_CASE1, PRESET_CASE2, PRESET_ROBER, PRESET_HYCHEM = 1, 2, 3, 4
SOLVER_ROSENBROCK23, SOLVER_TSIT5, SOLVER_AUTOTSIT5 = 0, 1, 2
GRAD_AUTO, GRAD_FORWARD, GRAD_ADJOINT = 0, 1, 2
QUEUE_AUTO, QUEUE_INDEX = 0, 1
JAC_ANALYTIC, JAC_FINITE_DIFF = 0, 1
CATH_SOLVER_ROSENBROCK23, CATH_SOLVER_AUTOTSIT5_TRBDF2, CATH_SOLVER_AUTOTSIT5_ROS23 = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRNN_HIP_LIB: load another build of the same ABI (kernel experiments, tools/); the default is the in-tree library
CSRC = os.path.join(_HERE, "csrc")
_OVERRIDE = os.environ.get("CRNN_HIP_LIB")
LIB_PATH = _OVERRIDE or os.path.join(CSRC, "libcrnn_hip.so")


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("ns", C.c_int32), ("nr", C.c_int32), ("has_temp", C.c_int32),
        ("param_map", C.c_int32), ("n_save", C.c_int32), ("loss_kind", C.c_int32), ("clamp_pred", C.c_int32),
        ("maxiters", C.c_int32), ("errnorm_sens", C.c_int32), ("device", C.c_int32), ("cols_per_lane", C.c_int32),
        ("solver", C.c_int32), ("grad_mode", C.c_int32), ("tape_steps", C.c_int32), ("rhs_kind", C.c_int32),
        ("lb", C.c_double), ("ub", C.c_double), ("inv_R", C.c_double), ("t0", C.c_double),
        ("atol", C.c_double * MAX_N), ("rtol", C.c_double * MAX_N), ("rate_scale", C.c_double * MAX_N),
        ("mw", C.c_double * MAX_N), ("gas_const", C.c_double),
        ("gamma", C.c_double), ("qmin", C.c_double), ("qmax", C.c_double), ("beta1", C.c_double),
        ("beta2", C.c_double), ("qsteady_min", C.c_double), ("qsteady_max", C.c_double),
        ("qoldinit", C.c_double), ("dtmin", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [("n_traj", C.c_int64), ("n_ok", C.c_int64), ("n_accept", C.c_int64), ("n_reject", C.c_int64),
                ("kernel_ms", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class CathodeConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("maxiters", C.c_int32), ("grad_mode", C.c_int32),
                ("lb_clamp", C.c_double), ("T0", C.c_double), ("atol", C.c_double), ("rtol", C.c_double),
                ("gamma", C.c_double), ("qmin", C.c_double), ("qmax", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("qsteady_min", C.c_double), ("qsteady_max", C.c_double), ("qoldinit", C.c_double)]


class OptConfig(C.Structure):
    _fields_ = [("use_expdecay", C.c_int32), ("decay_step", C.c_int32), ("ed_eta0", C.c_double),
                ("ed_decay", C.c_double), ("ed_clip", C.c_double), ("eta", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("wd", C.c_double), ("grad_clip_norm", C.c_double)]


# every symbol include/crnn_hip.h declares: name -> (restype, argtypes)
_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)
_CTX = C.c_void_p
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)   # crnn_allreduce_fn
SYMBOLS = {
    "crnn_abi_version": (C.c_int32, []),
    "crnn_build_info": (C.c_char_p, []),
    "crnn_debug_bounds": (C.c_int32, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "crnn_sizeof": (C.c_int32, [C.c_int32]),
    "crnn_last_error": (C.c_char_p, [_CTX]),
    "crnn_config_preset": (C.c_int32, [C.POINTER(Config), C.c_int32]),
    "crnn_opt_preset": (C.c_int32, [C.POINTER(OptConfig), C.c_int32]),
    "crnn_config_set_solver": (C.c_int32, [C.POINTER(Config), C.c_int32]),
    "crnn_n_params": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "crnn_n_theta": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "crnn_config_n_theta": (C.c_int32, [C.POINTER(Config)]),
    "crnn_p2vec": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, _DP, _DP, _DP]),
    "crnn_ctx_create": (C.c_int32, [C.POINTER(Config), C.POINTER(_CTX)]),
    "crnn_ctx_destroy": (None, [_CTX]),
    "crnn_ctx_set_stream": (C.c_int32, [_CTX, C.c_void_p]),
    "crnn_ctx_set_data": (C.c_int32, [_CTX, _DP, _DP, _DP, _DP, _IP, C.c_int32, C.c_int64]),
    "crnn_ctx_set_data_device": (C.c_int32, [_CTX, C.c_void_p, C.c_void_p, _DP, _DP, _IP, C.c_int32, C.c_int64]),
    "crnn_ctx_set_tables": (C.c_int32, [_CTX, _DP, _DP]),
    "crnn_solve": (C.c_int32, [_CTX, _DP, _DP, C.c_int32, C.c_int64, C.c_int64, C.c_int32, _DP, _DP, _DP, _IP, _IP,
                               C.POINTER(Stats)]),
    "crnn_loss_grad": (C.c_int32, [_CTX, _DP, C.c_int64, C.c_int64, C.c_int32, _DP, _DP, C.POINTER(Stats)]),
    "crnn_opt_state_len": (C.c_int32, [C.c_int32]),
    "crnn_opt_init": (C.c_int32, [C.POINTER(OptConfig), C.c_int32, _DP]),
    "crnn_opt_update": (C.c_int32, [C.POINTER(OptConfig), C.c_int32, _DP, _DP, _DP]),
    "crnn_train_init": (C.c_int32, [_CTX, C.POINTER(OptConfig), _DP]),
    "crnn_train_step": (C.c_int32, [_CTX, C.c_int64, C.c_int64, C.c_int32, _DP]),
    "crnn_train_step_begin": (C.c_int32, [_CTX, C.c_int64, C.c_int64, C.c_int32]),
    "crnn_train_step_end": (C.c_int32, [_CTX, _DP]),
    "crnn_grad_buffer": (C.c_int32, [_CTX, C.POINTER(C.c_void_p), _IP]),
    "crnn_get_params": (C.c_int32, [_CTX, _DP]),
    "crnn_set_params": (C.c_int32, [_CTX, _DP]),
    "crnn_get_opt_state": (C.c_int32, [_CTX, _DP]),
    "crnn_set_opt_state": (C.c_int32, [_CTX, _DP]),
    "crnn_train_update": (C.c_int32, [_CTX, _DP]),
    "crnn_last_stats": (C.c_int32, [_CTX, C.POINTER(Stats)]),
    "crnn_ctx_set_queue_order": (C.c_int32, [_CTX, C.c_int32]),
    "crnn_ctx_set_lanes_per_traj": (C.c_int32, [_CTX, C.c_int32]),
    "crnn_last_lanes_per_traj": (C.c_int32, [_CTX]),
    "crnn_tape_retries": (C.c_int64, [_CTX]),
    "crnn_hychem_block_cap": (C.c_int32, [_CTX]),
    "crnn_ctx_set_jacobian": (C.c_int32, [_CTX, C.c_int32]),
    "crnn_last_step_counts": (C.c_int32, [_CTX, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "crnn_kernel_times": (C.c_int32, [_CTX, _DP, C.c_int32]),
    "crnn_synchronize": (C.c_int32, [_CTX]),
    "crnn_comm_get_unique_id": (C.c_int32, [C.c_char_p]),
    "crnn_comm_init": (C.c_int32, [_CTX, C.c_char_p, C.c_int32, C.c_int32]),
    "crnn_comm_destroy": (C.c_int32, [_CTX]),
    "crnn_comm_set_allreduce": (C.c_int32, [_CTX, ALLREDUCE_FN, C.c_void_p]),
    "crnn_comm_collectives": (C.c_int64, [_CTX]),
    "crnn_allreduce_grad": (C.c_int32, [_CTX, _DP, C.c_int32]),
    "crnn_cathode_config_default": (C.c_int32, [C.POINTER(CathodeConfig)]),
    "crnn_cathode_create": (C.c_int32, [C.POINTER(CathodeConfig), C.POINTER(_CTX)]),
    "crnn_cathode_destroy": (None, [_CTX]),
    "crnn_cathode_last_error": (C.c_char_p, [_CTX]),
    "crnn_cathode_set_obs": (C.c_int32, [_CTX, C.c_int32, C.c_int32, _IP, _DP, _DP, _DP, _DP]),
    "crnn_cathode_solve": (C.c_int32, [_CTX, _DP, C.c_int64, _DP, _DP, _DP, _IP, _IP, C.POINTER(Stats)]),
    "crnn_cathode_comm_init": (C.c_int32, [_CTX, C.c_char_p, C.c_int32, C.c_int32]),
    "crnn_cathode_comm_destroy": (C.c_int32, [_CTX]),
    "crnn_cathode_allgather": (C.c_int32, [_CTX, _DP, C.c_int64, C.c_int32, C.c_int64, _DP]),
    "crnn_cathode_set_tape_every": (C.c_int32, [_CTX, C.c_int32]),
    "crnn_cathode_set_solver": (C.c_int32, [_CTX, C.c_int32]),
    "crnn_cathode_set_errnorm_sens": (C.c_int32, [_CTX, C.c_int32, _DP]),
    "crnn_cathode_last_chunk_stats": (C.c_int32, [_CTX, C.POINTER(C.c_int64)]),
    "crnn_cathode_set_particles": (C.c_int32, [_CTX, _DP, _DP, C.c_int64]),
    "crnn_cathode_svgd_step": (C.c_int32, [_CTX, C.c_int32, _DP, C.c_double, C.c_double, _DP, _DP, _DP]),
    "crnn_cathode_get_particles": (C.c_int32, [_CTX, _DP]),
    "crnn_svgd_update": (C.c_int32, [C.c_int32, _DP, _DP, C.c_int64, C.c_int32, C.c_double, C.c_double, _DP,
                                     C.POINTER(C.c_double), _DP, _DP]),
}


def source_hash() -> str:
    """sha256 prefix over the sorted csrc/*.hip, *.hpp, include/crnn_hip.h and csrc/Makefile (the default flags are part of
    what the binary is) -- the same digest the Makefile bakes into the binary (crnn_build_info) and writes to
    libcrnn_hip.so.srchash."""
    import hashlib
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(CSRC, "..", "..", "include", "crnn_hip.h"), "rb").read())
    h.update(open(os.path.join(CSRC, "Makefile"), "rb").read())
    return h.hexdigest()[:16]


def ensure_built(force: bool = False) -> str:
    """(Re)build libcrnn_hip.so when it is missing or was not compiled from the sources next to it: a binary that
    travelled with a snapshot must never run in place of the code it travelled with.  Returns the source hash.
    One process builds (file lock; torchrun ranks import at the same moment), into a temporary name that is moved into
    place, so that nobody maps a half-written library."""
    import fcntl
    import shutil
    import subprocess
    want = source_hash()
    side = LIB_PATH + ".srchash"

    def stale():
        have = open(side).read().strip() if os.path.exists(side) else None
        return (not os.path.exists(LIB_PATH) or have != want), have

    need, have = stale()
    if not (force or need):
        return want
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        raise ImportError(f"{LIB_PATH} is missing or stale (sources {want}, binary {have}) and hipcc was not found to "
                          "rebuild it. crnn_amd has no CPU fallback.")
    try:
        lock = open(os.path.join(CSRC, ".build.lock"), "w")
    except OSError as e:
        raise ImportError(f"{LIB_PATH} is missing or stale (sources {want}, binary {have}) and {CSRC} is not writable: "
                          "build it with `make -C crnn_amd/csrc` where it is") from e
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or stale()[0]:                      # re-read under the lock: another rank may have built it meanwhile
            tmp = f"libcrnn_hip.so.tmp{os.getpid()}"
            try:
                subprocess.check_call(["make", "-C", CSRC, "-B", "-s", f"OUT={tmp}"])
                os.replace(os.path.join(CSRC, tmp), LIB_PATH)
                os.replace(os.path.join(CSRC, tmp + ".srchash"), side)
            finally:
                for f in (tmp, tmp + ".srchash"):
                    if os.path.exists(os.path.join(CSRC, f)):
                        os.remove(os.path.join(CSRC, f))
    return want


def _load():
    want = None if _OVERRIDE else ensure_built()      # an explicit CRNN_HIP_LIB is the caller's responsibility
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    info = lib.crnn_build_info().decode()
    if want is not None and f"src={want} " not in info:
        raise ImportError(f"libcrnn_hip.so was built from other sources ({info}) than those next to it (src={want}); "
                          "rebuild with `make -C crnn_amd/csrc -B`")
    if lib.crnn_abi_version() != ABI_VERSION:
        raise ImportError(f"libcrnn_hip.so ABI {lib.crnn_abi_version()} != binding ABI {ABI_VERSION}")
    for which, cls in enumerate((Config, Stats, OptConfig, CathodeConfig)):
        if lib.crnn_sizeof(which) != C.sizeof(cls):
            raise ImportError(f"struct layout mismatch for {cls.__name__}: C {lib.crnn_sizeof(which)} != ctypes {C.sizeof(cls)}")
    return lib


lib = _load()


class CrnnError(RuntimeError):
    pass


def check(rc: int, ctx=None):
    if rc != 0:
        msg = lib.crnn_last_error(ctx)
        raise CrnnError(msg.decode() if msg else f"libcrnn_hip error {rc}")


def dptr(a):
    return a.ctypes.data_as(_DP) if a is not None else None


def iptr(a):
    return a.ctypes.data_as(_IP) if a is not None else None
