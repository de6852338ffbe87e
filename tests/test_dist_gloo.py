"""N > 1 path on CPU: two processes, gloo backend.

What is exercised is the product's data-parallel *host logic* (crnn_amd/dist.py:
contiguous sharding of the IC axis, one all-reduce of the
[grad_sum | n_overflow | loss_sum, n_ok, n_accept, n_reject, n_traj] vector per step,
identical replicated optimiser update on every rank).  There is no GPU here, so
each rank's shard result is produced by the CPU oracle in the exact layout
libcrnn_hip's reduction buffer has; the assertion is that two ranks reproduce
the single-process update bit-for-bit in the parameters they end with."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, oracle_problem


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_buffer(orc, setup, p, first, count, P, pad):
    th, dth = orc.p2vec(2, 6, 3, p)
    pb = oracle_problem(orc, "case2", setup)
    r = orc.solve_batch(pb, th, np.ascontiguousarray(setup["u0"].T), setup["tsteps"],
                        np.ascontiguousarray(setup["data"].transpose(2, 1, 0)), dtheta=dth, first=first, count=count)
    buf = np.zeros(P + pad + 5)
    buf[:P] = r["grad"]
    buf[-5] = r["loss"][first:first + count].sum()
    buf[-4] = float(np.sum(r["retcode"][first:first + count] == 0))
    buf[-3], buf[-2], buf[-1] = r["naccept"], r["nreject"], count
    return buf


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        import json
        from crnn_amd import Optimiser, PRESET_CASE2
        from crnn_amd.dist import allreduce_sum_, mean_loss_and_grad_from_sums, shard_range
        from oracle import oracle as orc
        fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
        c2 = fx["case2"]
        setup = dict(u0=np.array(c2["u0"]), tsteps=np.array(c2["tsteps"]), data=np.array(c2["data"]),
                     yscale=np.array(c2["yscale"]))
        p = np.array(c2["p_init"])
        opt = Optimiser(25, PRESET_CASE2)
        B = setup["u0"].shape[0]
        first, count = shard_range(B, rank, world)
        losses = []
        for _ in range(3):
            buf = _shard_buffer(orc, setup, p, first, count, 25, 1)
            allreduce_sum_(buf)
            loss, grad = mean_loss_and_grad_from_sums(buf, 25)
            opt.update_(p, grad)
            losses.append(loss)
        q.put((rank, p.copy(), losses))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_data_parallel_equals_single_process(orc, case2_setup):
    from crnn_amd import Optimiser, PRESET_CASE2
    from crnn_amd.dist import mean_loss_and_grad_from_sums
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for pr in procs:
        pr.start()
    try:
        res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
        for pr in procs:
            pr.join(60)
            assert pr.exitcode == 0
    finally:
        for pr in procs:
            if pr.is_alive():
                pr.kill()
                pr.join(30)
    # single-process reference (whole ensemble, same oracle stand-in)
    p = case2_setup["p_init"].copy()
    opt = Optimiser(25, PRESET_CASE2)
    B = case2_setup["u0"].shape[0]
    ref_losses = []
    for _ in range(3):
        buf = _shard_buffer(orc, case2_setup, p, 0, B, 25, 1)
        loss, grad = mean_loss_and_grad_from_sums(buf, 25)
        opt.update_(p, grad)
        ref_losses.append(loss)
    assert np.array_equal(res[0][1], res[1][1]), "replicated optimiser states diverged between ranks"
    # the two-rank sum differs from the one-rank sum only by floating-point association
    assert np.max(np.abs(res[0][1] - p)) < 1e-12
    assert np.allclose(res[0][2], ref_losses, rtol=1e-12, atol=0)


def _svgd_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        from crnn_amd.dist import allgather_rows, shard_range
        from oracle.oracle import svgd_update      # CPU stand-in for the device SVGD move (no GPU here)
        rng = np.random.default_rng(0)
        N = 11                                       # not divisible by the world size
        p = 1 + 0.05 * rng.standard_normal((N, 17))
        for _ in range(3):
            first, count = shard_range(N, rank, world)
            local = np.sin(p[first:first + count] * 3.0) - 0.1 * p[first:first + count]     # stand-in for the GPU's lnpgrad rows
            lnpgrad = allgather_rows(local, N)
            p = svgd_update(p, lnpgrad, 0.05)[0]
        q.put((rank, p))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_particle_sharding_and_svgd_update():
    """Cathode-UQ exchange: particles sharded, one all-gather of the gradient rows, replicated SVGD move."""
    from oracle.oracle import svgd_update
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_svgd_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for pr in procs:
        pr.start()
    try:
        res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
        for pr in procs:
            pr.join(60)
            assert pr.exitcode == 0
    finally:
        for pr in procs:
            if pr.is_alive():
                pr.kill()
                pr.join(30)
    rng = np.random.default_rng(0)
    p = 1 + 0.05 * rng.standard_normal((11, 17))
    for _ in range(3):
        p = svgd_update(p, np.sin(p * 3.0) - 0.1 * p, 0.05)[0]
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][1], p)
