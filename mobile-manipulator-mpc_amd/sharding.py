"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed).

Problem instances are independent (controllers/mpc_wholebody_qref.py:287-331 touches only its own
x_init / traj_ref / u_ref / u_latest), so rank r owns the contiguous slice [lo, hi) of the global batch
and solves it with no data-path collective.  The only exchange is the all-gather of the solved
trajectories (X, U, s) - RCCL over xGMI with backend "nccl", gloo on CPU for tests.
"""
import torch


def shard_bounds(B, world, rank):
    """Contiguous, balanced slices: sizes differ by at most one."""
    return rank * B // world, (rank + 1) * B // world


def record_len(N, nx, nu):
    return (N + 1) * nx + N * nu + (N + 1)


def pack_solution(X, U, s, out=None):
    """(B,N+1,nx), (B,N,nu), (B,N+1) -> (B, record_len) rows [X | U | s]."""
    B = X.shape[0]
    a, b = X[0].numel(), U[0].numel()
    if out is None:
        out = torch.empty((B, a + b + s.shape[1]), dtype=X.dtype, device=X.device)
    out[:, :a] = X.reshape(B, -1)
    out[:, a:a + b] = U.reshape(B, -1)
    out[:, a + b:] = s
    return out


def unpack_solution(rec, N, nx, nu):
    B = rec.shape[0]
    a, b = (N + 1) * nx, N * nu
    return rec[:, :a].reshape(B, N + 1, nx), rec[:, a:a + b].reshape(B, N, nu), rec[:, a + b:]


def allgather_solutions(packed_local, B_global, dist=None, gathered=None, async_op=False):
    """All ranks end up with the (B_global, record_len) table in global instance order.
    Equal shards use one all_gather_into_tensor; ragged shards fall back to all_gather of padded rows.
    async_op=True (equal shards only) returns (gathered, work): the collective runs on the backend's own stream and the
    caller waits on `work` before it reads `gathered` or reuses `packed_local` - the next batch can be solved meanwhile."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return (packed_local, None) if async_op else packed_local
    world, rank = dist.get_world_size(), dist.get_rank()
    rec = packed_local.shape[1]
    if gathered is None:
        gathered = torch.empty((B_global, rec), dtype=packed_local.dtype, device=packed_local.device)
    sizes = [shard_bounds(B_global, world, r)[1] - shard_bounds(B_global, world, r)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        work = dist.all_gather_into_tensor(gathered, packed_local.contiguous(), async_op=async_op)
        return (gathered, work) if async_op else gathered
    mx = max(sizes)
    pad = torch.zeros((mx, rec), dtype=packed_local.dtype, device=packed_local.device)
    pad[:packed_local.shape[0]] = packed_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    o = 0
    for r in range(world):
        gathered[o:o + sizes[r]] = parts[r][:sizes[r]]
        o += sizes[r]
    return (gathered, None) if async_op else gathered
