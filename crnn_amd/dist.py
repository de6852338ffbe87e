"""Data-parallel sharding of the ensemble of initial conditions.

The reference has no parallelism at all; the only natural shard
axis of the hot path is the IC axis: trajectories are independent given the
weights.  One process per GPU; each rank owns a contiguous block of ICs that
never moves; per optimiser step the ranks exchange exactly one small vector
    [ sum_b d loss_b / d p  |  n_overflow  | loss_sum, n_ok, n_accept, n_reject, n_traj ]      (P + 6 doubles)
with one all-reduce (RCCL over xGMI: in-library ncclAllReduce on the ctx stream,
torch.distributed on the same buffer, or a caller-supplied collective) and then
apply the identical, deterministic optimiser update on every rank (no broadcast
needed).  The layout is the same whichever gradient algorithm a rank used, so a
rank that fell back from the adjoint to forward tangents still sums correctly;
n_overflow != 0 after the sum makes EVERY rank skip the step and replay it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L
from ._lib import check, lib


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous block [first, first+count) of rank's share of n_total items
    (the first n_total % world ranks get one extra)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def allreduce_sum_(buf: np.ndarray, group=None) -> np.ndarray:
    """In-place sum of a host float64 vector over all ranks of the default torch.distributed
    group (gloo on CPU).  Identity when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return buf
    t = torch.from_numpy(buf)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return buf


def mean_loss_and_grad_from_sums(buf: np.ndarray, n_params: int):
    """Unpack an all-reduced [grad_sum | ... | loss_sum, n_ok, n_accept, n_reject, n_traj] vector."""
    n_traj = buf[-1]
    if n_traj <= 0:
        raise ValueError("no trajectories contributed")
    return buf[-5] / n_traj, buf[:n_params] / n_traj


class _DevView:
    """Zero-copy __cuda_array_interface__ view of a device double buffer (for torch.as_tensor)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class DataParallel:
    """Attach a NeuralODE (one per rank) to the other ranks.

    comm="rccl": the library's own communicator (crnn_comm_init; the unique id
    is broadcast through torch.distributed).  comm="torch": torch.distributed
    all_reduce (backend nccl == RCCL) on the library's gradient buffer between
    crnn_train_step_begin / _end, with the ctx bound to torch's current stream.
    comm="callback": the fused crnn_train_step (steps enqueued back to back, the
    tape-overflow outcome looked at later) with torch.distributed's all_reduce
    handed to the library as its collective (crnn_comm_set_allreduce); works on
    any backend -- on gloo the 248-byte vector goes through the host.
    """

    def __init__(self, node, comm: str = "rccl"):
        import torch
        import torch.distributed as dist
        self.node, self.comm = node, comm
        self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        self._torch, self._dist = torch, dist
        if comm == "rccl":
            uid = C.create_string_buffer(L.UNIQUE_ID_BYTES)
            if self.rank == 0:
                check(lib.crnn_comm_get_unique_id(uid))
            if self.world > 1:
                obj = [uid.raw]
                dist.broadcast_object_list(obj, src=0)
                uid = C.create_string_buffer(obj[0], L.UNIQUE_ID_BYTES)
            check(lib.crnn_comm_init(node.handle, uid, self.rank, self.world), node.handle)
        elif comm in ("torch", "callback"):
            stream = torch.cuda.current_stream().cuda_stream
            check(lib.crnn_ctx_set_stream(node.handle, C.c_void_p(stream)), node.handle)
            if comm == "callback":
                on_host = dist.is_initialized() and dist.get_backend() == "gloo"

                def _allreduce(d_buf, n, _stream, _user):
                    try:
                        if self.world > 1:
                            t = torch.as_tensor(_DevView(d_buf, n), device="cuda")
                            if on_host:
                                h = t.cpu()                       # orders after the solve on the (shared) current stream
                                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                                t.copy_(h)
                            else:
                                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                        return 0
                    except Exception as exc:                      # never unwind through the C frame
                        print(f"crnn_amd.dist: all-reduce callback failed: {exc!r}", flush=True)
                        return 1

                self._cb = L.ALLREDUCE_FN(_allreduce)             # keep the trampoline alive as long as the ctx may call it
                check(lib.crnn_comm_set_allreduce(node.handle, self._cb, None), node.handle)
        else:
            raise ValueError("comm must be 'rccl', 'torch' or 'callback'")

    def train_step(self, first=0, count=None, sample=None, want_loss=False):
        node = self.node
        if self.comm in ("rccl", "callback"):
            return node.train_step(first, count, sample, want_loss)
        count = node.B - first if count is None else count
        sample = node.D if sample is None else int(sample)
        check(lib.crnn_train_step_begin(node.handle, first, count, sample), node.handle)
        if self.world > 1:
            ptr, n = C.c_void_p(), C.c_int32()
            check(lib.crnn_grad_buffer(node.handle, C.byref(ptr), C.byref(n)), node.handle)
            t = self._torch.as_tensor(_DevView(ptr.value, n.value), device="cuda")
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        loss = C.c_double(0.0)
        check(lib.crnn_train_step_end(node.handle, C.byref(loss) if want_loss else None), node.handle)
        return loss.value if want_loss else None

    def selftest(self) -> bool:
        """All-reduce a known vector through the chosen communicator and check the sum."""
        n = 27
        mine = np.arange(n, dtype=np.float64) * (self.rank + 1)
        want = np.arange(n, dtype=np.float64) * (self.world * (self.world + 1) / 2)
        if self.comm == "rccl":
            buf = mine.copy()
            check(lib.crnn_allreduce_grad(self.node.handle, L.dptr(buf), n), self.node.handle)
            return bool(np.array_equal(buf, want))
        t = self._torch.tensor(mine, device="cuda")
        if self.world > 1:
            self._dist.all_reduce(t)
        return bool(np.array_equal(t.cpu().numpy(), want))

    def collectives(self) -> int:
        """All-reduces the library's training loop has issued on this rank (rccl / callback modes)."""
        return int(lib.crnn_comm_collectives(self.node.handle))

    def close(self):
        if self.comm == "rccl":
            lib.crnn_comm_destroy(self.node.handle)
        elif self.comm == "callback":
            lib.crnn_comm_set_allreduce(self.node.handle, L.ALLREDUCE_FN(), None)


def allgather_rows(local: np.ndarray, n_total: int, group=None) -> np.ndarray:
    """Concatenate per-rank row blocks (rank r holds rows shard_range(n_total, r, world)) into the full [n_total, ...]
    array on every rank: the exchange step of the Bayesian cathode ensemble (particles are sharded, every rank then
    forms the full SVGD update).  Identity when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if local.shape[0] != n_total:
            raise ValueError("single process must hold all rows")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, count = shard_range(n_total, rank, world)
    if local.shape[0] != count:
        raise ValueError(f"rank {rank} must hold {count} rows, got {local.shape[0]}")
    cmax = -(-n_total // world)
    pad = np.zeros((cmax,) + local.shape[1:], dtype=np.float64)
    pad[:count] = local
    # nccl (== RCCL) moves device tensors only; gloo moves host tensors
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    send = torch.from_numpy(pad).to(dev)
    outs = [torch.zeros(pad.shape, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(outs, send, group=group)
    full = np.empty((n_total,) + local.shape[1:], dtype=np.float64)
    for r in range(world):
        f, c = shard_range(n_total, r, world)
        full[f:f + c] = outs[r].cpu().numpy()[:c]
    return full
