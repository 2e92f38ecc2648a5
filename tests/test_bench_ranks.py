"""GPU box: bench.py's multi-rank path (per-rank seeded batches, overlapped all-gather of the solutions, max-over-ranks
timing, one JSON line from rank 0) rehearsed with two ranks that share the one GPU - the all-gather goes through gloo
(MMPC_BENCH_BACKEND), the solves through the HIP library.  The driver's own N-GPU runs use RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_two_rank_strong_scaling_u0_gather():
    """--scaling strong (global batch fixed, contiguous slices of one seeded batch) with the u0-only all-gather."""
    env = dict(os.environ, MMPC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2048", "--scaling", "strong", "--gather", "u0"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["batch_per_gpu"] == 1024 and d["config"]["seeds"] == [3]
    assert "all-gather(u0)" in d["config"]["parallelism"] and [p_["batch"] for p_ in d["per_rank"]] == [1024, 1024]
    assert d["solver"]["converged_frac"] == 1.0 and abs(d["value"] - 2048 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_two_rank_bench_line():
    env = dict(os.environ, MMPC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "1024"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["unit"] == "solves/s"
    assert d["config"]["seeds"] == [3, 4] and d["config"]["batch_per_gpu"] == 1024      # consecutive seeds, nothing filtered
    assert d["config"]["iteration_budget"] == 0                                          # the same mode at every N: one launch per batch, plain kernel
    assert d["gather_checked"] is True                                                   # every rank found its records in the gathered table
    pb, ss = d["pipelined_budget"], d["strong_scaling"]                                  # collective extras: pipelined budget, C4 read literally
    assert d["value_strong"] == ss["value"] and "value_strong" in d["config"]["workload"]  # the literal C4 figure at top level beside the weak one
    assert pb["iteration_budget"] == 64 and pb["converged_frac"] == 1.0 and pb["gather_checked"] is True and pb["value"] > 0
    assert ss["global_batch"] == 1024 and ss["batch_per_gpu"] == [512, 512] and ss["converged_frac"] == 1.0 and ss["gather_checked"] is True
    assert len(d["per_rank"]) == 2 and all(p["converged"] == 1024 for p in d["per_rank"]) and "no schedule hint from earlier solves" in d["config"]["workload"]
    assert d["solver"]["converged_frac"] == 1.0 and d["solver"]["max_scaled_kkt"] <= 1e-8
    assert d["value"] > 0 and abs(d["value"] - 2 * 1024 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d                       # rank 0 at N=1 only


@pytest.mark.gpu
def test_two_rank_c5_receding_horizon():
    """--config c5 under torchrun: per-rank device and robots (weak: seeds 5, 6), slice of u_latest resident per rank, the
    all-gather of u0 once per tick (gloo here, RCCL on a multi-GPU node)."""
    env = dict(os.environ, MMPC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "2",
           "--batch", "512", "--config", "c5", "--ticks", "3"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["batch_per_gpu"] == 512 and "all-gather(u0) per tick" in d["config"]["parallelism"]
    assert d["gather_checked"] is True and len(d["per_rank"]) == 2 and all(p_["converged"] == p_["solves"] == 512 * 3 for p_ in d["per_rank"])
    assert d["solver"]["converged_frac"] == 1.0 and abs(d["value"] - 2 * 512 * 3 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d
    # collective extra: every rank's robots as three groups out of phase, u0 gathered per group and tick on the group's stream;
    # per-robot results bitwise those of lock step, every rank's rows found in the gathered tables
    gr = d["groups_per_rank"]
    assert gr["groups"] == 3 and gr["checked"] is True and gr["value"] > 0 and "per group" in gr["gather"]
