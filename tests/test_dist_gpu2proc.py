"""N = 2 on ONE GPU: two processes share device 0, process group on gloo (RCCL refuses two ranks on one device; the
driver's 8-GPU run is the only place the in-library ncclAllReduce sees N > 1).  What runs here for real, on the device:
the library's training loop on two ranks with the ensemble sharded over the IC axis, the exchange of the
[grad | n_overflow | loss_sum, n_ok, n_accept, n_reject, n_traj] vector, and -- the case the one-rank tests cannot
reach -- a tape overflow on ONE rank only:

  comm="callback"  crnn_train_step enqueues steps without looking (deferred); rank 1's overflow count travels through the
                   all-reduce, BOTH ranks skip the step (and everything after it) and BOTH replay the skipped steps with
                   forward tangents when the host looks: same number of collectives on both ranks, parameters bit-identical
                   across ranks and bit-identical to a run that used forward tangents from the start;
  comm="torch"     crnn_train_step_begin / _end: rank 1 falls back to forward tangents locally, the reduced vector keeps its
                   one layout (P + 6 doubles whatever algorithm produced it), so the element-wise sum stays meaningful
                   (round 1's forward path had a different length and offset: ADVICE r1).
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, mode, tape1, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import json

    import torch
    import torch.distributed as dist
    import datetime
    torch.cuda.set_device(0)
    # a collective whose partner died must fail, not wait for ever
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_ROBER
        from crnn_amd.dist import DataParallel, shard_range
        fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
        rb = fx["robertson"]
        u0, ts, data, ys = np.array(rb["u0"]), np.array(rb["tsteps"]), np.array(rb["data"]), np.array(rb["yscale"])
        p0 = np.array(fx["rober_ckpt"]["p"])
        first, count = shard_range(u0.shape[0], rank, world)
        if mode.endswith("forward") or (mode.endswith("mixed") and rank == 0):
            kw = dict(grad_mode=1)     # "mixed": rank 0 on forward tangents (what grad_mode AUTO picks for a small LOCAL shard)
        else:
            kw = dict(grad_mode=2, tape_steps=(tape1 if rank == 1 else 0))
        node = NeuralODE(ODEProblem(PRESET_ROBER, ts, rate_scale=np.array(rb["dydt_scale"]), **kw))
        node.set_ensemble(u0[first:first + count], data[first:first + count], ys)      # this rank's shard only
        node.train_init(Optimiser(43, PRESET_ROBER), p0)
        dp = DataParallel(node, comm=mode.split("-")[0])
        for sm in (20, 40, 22, 40, 25, 40):
            dp.train_step(sample=sm, want_loss=False)
        p = node.params()
        q.put((rank, p, dp.collectives(), node.stats()["n_traj"]))
        dp.close()
        node.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(mode, tape1):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()                                    # a fresh port per run (a reused one may still be in TIME_WAIT)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, tape1, q), daemon=True) for r in range(2)]
    for pr in procs:
        pr.start()
    try:
        res = sorted([q.get(timeout=400) for _ in procs], key=lambda t: t[0])
        for pr in procs:
            pr.join(120)
            assert pr.exitcode == 0
    finally:
        for pr in procs:          # never leave a rank behind (it would block the interpreter's exit)
            if pr.is_alive():
                pr.kill()
                pr.join(30)
    return res


@pytest.mark.timeout(900)
def test_one_rank_overflows_deferred_skip_and_replay_stay_in_lockstep():
    ref = _run("callback-forward", 0)              # forward tangents on both ranks from the start
    assert np.array_equal(ref[0][1], ref[1][1]) and ref[0][2] == ref[1][2] == 6
    roomy = _run("callback", 0)                    # adjoint, nobody overflows: 6 collectives, gradients equal to rounding
    assert np.array_equal(roomy[0][1], roomy[1][1]) and roomy[0][2] == roomy[1][2] == 6
    assert np.max(np.abs(roomy[0][1] - ref[0][1])) < 1e-9
    # rank 1 records 30 steps per lane: horizons 20-25 fit, the full horizon (about 35 accepted steps) overflows there,
    # never on rank 0.  Step 2 is skipped by BOTH ranks, so are steps 3..6 (sticky); all five are replayed in order.
    tiny = _run("callback", 30)
    assert np.array_equal(tiny[0][1], tiny[1][1]), "the ranks' replicated parameters diverged"
    assert tiny[0][2] == tiny[1][2] == 6 + 5, (tiny[0][2], tiny[1][2])
    assert np.max(np.abs(tiny[0][1] - ref[0][1])) < 1e-9          # step 1 by the adjoint, steps 2-6 by forward tangents
    # every step overflows on rank 1: the whole run is replayed with forward tangents -> bit-identical to `ref`
    allo = _run("callback", 4)
    assert np.array_equal(allo[0][1], allo[1][1]) and allo[0][2] == allo[1][2] == 12
    assert np.array_equal(allo[0][1], ref[0][1])


@pytest.mark.timeout(900)
def test_forward_rank_replays_with_the_overflowing_adjoint_rank():
    """the ranks need not run the same algorithm (grad_mode AUTO picks forward tangents from the LOCAL count).
    Rank 0 on forward tangents never defers anything itself, but the summed overflow count makes its optimiser kernel skip
    the step like rank 1's: it must look at the sticky flag and replay too, or the collectives pair up wrongly."""
    ref = _run("callback-forward", 0)
    mixed = _run("callback-mixed", 30)             # rank 1: adjoint, tape of 30 steps -> overflows on the full horizon
    assert np.array_equal(mixed[0][1], mixed[1][1]), "the ranks' replicated parameters diverged"
    assert mixed[0][2] == mixed[1][2] == 6 + 5, (mixed[0][2], mixed[1][2])
    assert np.max(np.abs(mixed[0][1] - ref[0][1])) < 1e-9


@pytest.mark.timeout(900)
def test_one_rank_falls_back_locally_in_the_split_api():
    ref = _run("torch-forward", 0)
    mix = _run("torch", 4)                          # rank 0: adjoint; rank 1: tape of 4 steps -> forward tangents, every step
    assert np.array_equal(mix[0][1], mix[1][1]), "the ranks summed buffers of different layout"
    assert np.all(np.isfinite(mix[0][1])) and np.max(np.abs(mix[0][1] - ref[0][1])) < 1e-9
