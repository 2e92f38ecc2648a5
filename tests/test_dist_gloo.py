"""CPU: the multi-GPU path (batch sharding + the one all-gather of solved trajectories) with world_size 2 on
gloo.  The per-rank 'solver' here is the CPU oracle on the rank's slice - what is under test is the sharding /
packing / gather logic that bench.py and the N-GPU driver use, not the HIP kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mmpc_loader
    from oracle import synth, coracle, nlp
    mm = mmpc_loader.load()
    from mmpc_amd import sharding
    N, nx, nu = 20, 9, 5
    d = synth.make_batch(B)
    lo, hi = sharding.shard_bounds(B, world, rank)
    r = coracle.solve_batch(nlp.WholeBodyParams(), d["x_init"][lo:hi], d["traj_ref"][lo:hi], d["u_ref"][lo:hi],
                            np.zeros((hi - lo, N, nu)), d["obs"][lo:hi])
    packed = sharding.pack_solution(torch.from_numpy(r["X"]), torch.from_numpy(r["U"]), torch.from_numpy(r["s"]))
    table = sharding.allgather_solutions(packed, B, dist)
    # the asynchronous form bench.py uses (double-buffered: the gather travels while the next batch is solved)
    table2, work = sharding.allgather_solutions(packed, B, dist, async_op=True)
    if work is not None:
        work.wait()
    assert torch.equal(table2, table)
    X, U, s = sharding.unpack_solution(table, N, nx, nu)
    q.put((rank, X.numpy().copy(), U.numpy().copy(), s.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])       # equal and ragged shards
def test_two_rank_shard_and_allgather(B):
    from oracle import synth, coracle, nlp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + B
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = synth.make_batch(B)
    full = coracle.solve_batch(nlp.WholeBodyParams(), d["x_init"], d["traj_ref"], d["u_ref"], np.zeros((B, 20, 5)), d["obs"])
    for rank, X, U, s in got:
        assert np.array_equal(X, full["X"]) and np.array_equal(U, full["U"]) and np.array_equal(s, full["s"])


def test_shard_bounds_cover_batch():
    sys.path[:0] = [ROOT]
    import mmpc_loader
    mmpc_loader.load()
    from mmpc_amd import sharding
    for B in (1, 7, 8192, 8191):
        for W in (1, 2, 3, 8):
            b = [sharding.shard_bounds(B, W, r) for r in range(W)]
            assert b[0][0] == 0 and b[-1][1] == B and all(b[i][1] == b[i + 1][0] for i in range(W - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
