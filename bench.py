"""MPC solves/sec, whole-body N=20, batch 8192 (BASELINE.json metric) on 1/2/4/8 MI355X.

One "step" = one pass of the hot path (MPCWholeBody.solve for every instance of the batch,
controllers/mpc_wholebody_qref.py:287-331) over a fixed synthetic batch whose inputs are already
resident in HBM.  N ranks: the global batch is sharded contiguously over ranks (no data-path
collective), followed by the one all-gather of the solved trajectories the metric's config names.
Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
_nullcontext = contextlib.nullcontext   # (main() has a local `import contextlib` further down: nested functions use this name)
import json
import os
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = FP64 matrix peak (spec), SURVEY §7
HBM_PEAK_GBS = 8000.0
PROFILE_TAG = "r04"
SEED_BASE = 3                                     # weak scaling: rank r solves seed SEED_BASE + r of the C4 generator (no selection)



def _dig(d, path):
    """d[path[0]][path[1]]... or None where a key is missing (extras that did not run)"""
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d if isinstance(d, (int, float)) else None

def _synth():
    """the package's seeded workload generator (numpy only): mobile-manipulator-mpc_amd/synth.py"""
    import mmpc_loader
    return mmpc_loader.load().synth

def riccati_flops_per_iter(N, nx, nu, M, nself):
    """Algorithmic flops of ONE interior-point iteration of the stage-wise (Riccati) KKT
    factorisation, dense-block convention: per stage
    P[A B] 2nx^2(nx+nu) + [A B]^T(.) 2(nx+nu)^2 nx + chol nu^3/3 + solves 2nu^2(nx+1) + P update 2nx^2 nu,
    plus ~ (60 M + 250 nself + 8 (nx+nu)) per stage for row evaluation / barrier terms, twice
    (direction + one line-search trial)."""
    nv = nx + nu
    ric = 2 * nx * nx * nv + 2 * nv * nv * nx + nu ** 3 / 3.0 + 2 * nu * nu * (nx + 1) + 2 * nx * nx * nu
    rows = 2 * (60 * M + 250 * nself + 8 * nv)
    return N * ric + (N + 1) * rows


def algorithmic_bytes_per_solve(N, nx, nu, M):
    """compulsory fp64 read-once/write-once traffic per solve."""
    return 8 * (nx + nx * (N + 1) + nu * N + nu * N + 3 * M) + 8 * (nx * (N + 1) + nu * N + (N + 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8192, help="batch of the BASELINE metric (8192)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank solves `--batch` instances (global = batch x N), no data-path collective "
                         "needed by the solve itself; strong: the global batch stays `--batch` (BASELINE config C4 read literally)")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--obstacles", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=8192, help="instances timed on the host cores (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--seed-base", type=int, default=SEED_BASE, help="weak scaling: rank r solves the C4 seed SEED_BASE + r")
    ap.add_argument("--budget", type=int, default=0,
                    help="iterations per launch of the TIMED steps (mmpc_set_iteration_budget); instances that need more are finished by a "
                         "continuation launch on a side stream while the next batches run (--handles handles in rotation).  Default 0 "
                         "(off) at every N: one launch per batch on the plain kernel - the same mode on one GPU and on eight.  The "
                         "pipelined mode is measured beside it on the same batches (extra `pipelined_budget`, --pipelined-budget)")
    ap.add_argument("--pipelined-budget", type=int, default=64, help="iteration budget of the `pipelined_budget` extra")
    ap.add_argument("--no-extras", action="store_true", help="headline only (no collective extras, no rank-0 extras)")
    ap.add_argument("--handles", type=int, default=8, help="handles in rotation when --budget > 0")
    ap.add_argument("--side-streams", type=int, default=6, help="streams the continuation launches rotate over when --budget > 0")
    ap.add_argument("--gather", default="full", choices=["full", "u0"],
                    help="what the one all-gather of the path collects: the solved (X,U,s) records (310 doubles per instance) "
                         "or only the first inputs u0 the closed loop applies (5 doubles per instance, SURVEY 8e)")
    ap.add_argument("--config", default="c4", choices=["c4", "c5"],
                    help="c4: BASELINE metric (default). c5: N=30, 8 moving obstacles, warm-started receding horizon (1 GPU)")
    ap.add_argument("--ticks", type=int, default=10)
    ap.add_argument("--async-budget", type=int, default=192, help="--config c5: iterations per launch of the asynchronous receding-horizon extra")
    args = ap.parse_args()
    if args.config == "c5":
        return main_c5(args)
    if args.budget < 0 or args.pipelined_budget < 1:
        raise SystemExit("--budget must be >= 0, --pipelined-budget >= 1")

    # Pipelined continuation (see --budget): the launch stream plus the side streams of the continuations must not share a
    # hardware queue - a queue runs its kernels one after the other, the next batch would wait behind a 20 ms continuation.
    # ROCm's default is 4 queues per process; read at runtime initialisation, hence before torch is imported.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # (also for the one-GPU run: its extras time the same pattern)

    import torch
    import mmpc_loader
    mm = mmpc_loader.load()
    synth = _synth()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # MMPC_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # the all-gather goes through gloo); the driver's runs use the default, RCCL
    backend = os.environ.get("MMPC_BENCH_BACKEND", "nccl")
    local_dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    N, M = args.horizon, args.obstacles
    nx, nu = 9, 5
    from mmpc_amd import sharding
    robot = mm.MobileManipulator(0.1)

    def run_config(scaling, budget, steps, warmup):
        """Warm-up + `steps` timed passes of the hot path in one mode: `scaling` weak (every rank its own seeded batch) or strong
        (one seeded batch of --batch instances, contiguous slices), `budget` 0 (one launch per batch, the plain kernel) or > 0
        (at most `budget` iterations per launch, the rest by continuation launches on side streams while the next batches run).
        Collective: every rank calls it with the same arguments.  Timing per the contract: barrier + synchronize on both sides,
        max over ranks."""
        Bg = args.batch * (world if scaling == "weak" else 1)       # global number of instances
        lo, hi = sharding.shard_bounds(Bg, world, rank)
        Bl = hi - lo
        if scaling == "weak":
            # every rank generates its own seeded batch (rank 0 = the N=1 workload); instance b of rank r is global b + r*batch.
            # Consecutive seeds, nothing filtered: what the slowest instance of a rank's batch costs shows in per_rank below
            d = synth.make_batch(args.batch, N=N, M=M, config_id=args.seed_base + rank)
            sl = slice(0, Bl)
        else:
            # identical seeded inputs on every rank; each rank keeps its contiguous slice
            d = synth.make_batch(Bg, N=N, M=M, config_id=args.seed_base)
            sl = slice(lo, hi)
        ctrl = mm.MPCWholeBody(robot, [], [], N=N, max_batch=max(Bl, 1), device=local_dev, n_obstacles=M)
        eng = ctrl._engine
        # The headline uses NO information from earlier solves: the timed steps re-solve one resident batch, and a launch order
        # taken from the previous solve of the identical problems is knowledge no first solve of a batch has.  Mode 2 = the
        # engine orders the workgroups by an a-priori difficulty key computed from this batch's own data inside every call
        # (mmpc_set_schedule_hint; the key kernel and the sort are part of the timed launch)
        eng.set_schedule_hint(2)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl])).to(dev)
        x_init = t(np.clip(d["x_init"], ctrl.xlim[0], ctrl.xlim[1]))
        traj, uref, obs = t(d["traj_ref"]), t(d["u_ref"]), t(d["obs"])
        ulast = torch.zeros((Bl, N, nu), dtype=torch.float64, device=dev)   # cold start: u_latest = 0 (:298-299)
        out = eng.solve_batch_device(x_init, traj, uref, ulast, obs)
        # Iteration budget + continuation: a launch gives every instance at most `budget` iterations; the few that need more park
        # their state and are finished by a continuation launch on a side stream while the next batches run.  The handle's state
        # (save area, launch lists) belongs to one batch in flight, so NH handles take turns; a handle's next launch is ordered
        # after its continuation by the engine itself (event across streams).  Results are bitwise those of one uninterrupted
        # solve.  A batch takes as long as its slowest instance - a 450-iteration straggler is 16 ms against 6.5 ms - but the
        # throughput of a stream of batches no longer does.
        NH = args.handles if budget > 0 else 1
        engs, outs, sides, keep = [eng], [out], [None], [ctrl]
        if budget > 0:
            for _ in range(NH - 1):
                c_ = mm.MPCWholeBody(robot, [], [], N=N, max_batch=max(Bl, 1), device=local_dev, n_obstacles=M)
                c_._engine.set_schedule_hint(2)
                engs.append(c_._engine); outs.append(None); keep.append(c_)
            # (fewer side streams than hardware queues, see GPU_MAX_HW_QUEUES above; continuations on one stream run one after the other)
            sides = [torch.cuda.Stream(device=dev) for _ in range(min(NH, args.side_streams))]
            for h in range(NH):
                engs[h].set_iteration_budget(budget)
        packed = gathered = None
        pending = [None] * max(2, NH)
        rec = sharding.record_len(N, nx, nu) if args.gather == "full" else nu
        if world > 1:
            packed = [torch.empty((Bl, rec), dtype=torch.float64, device=dev) for _ in range(max(2, NH))]
            gathered = [torch.empty((Bg, rec), dtype=torch.float64, device=dev) for _ in range(max(2, NH))]

        def solve_step(i):
            # one pass of the hot path over the resident batch; returns the output set it writes and the stream its last launch is on
            if budget == 0:
                eng.solve_batch_device(x_init, traj, uref, ulast, obs, out=out)
                return out, None
            h = i % NH
            # the output set of this handle is still being packed / gathered from its previous turn (on a side stream, behind the
            # continuation): the launch that overwrites it is ordered after that gather
            if pending[h % len(pending)] is not None:
                pending[h % len(pending)].wait()
                pending[h % len(pending)] = None
            outs[h] = engs[h].solve_batch_device(x_init, traj, uref, ulast, obs, out=outs[h])
            side = sides[i % len(sides)]
            engs[h].resume_batch_device(x_init, traj, uref, ulast, obs, outs[h], stream=side.cuda_stream)
            return outs[h], side

        def gather(i, o, side):
            # the one collective of the path: all-gather of the solved (X,U,s) - or of u0 only - over xGMI (RCCL), inside the
            # timed region.  Multi-buffered and asynchronous: the gather of step i travels on RCCL's stream while step i+1 is
            # solved; a buffer set is reused only after its previous gather has completed, and drain() waits for the last ones.
            # With a continuation in flight the packing and the gather are issued on that continuation's stream (behind it).
            b = i % len(pending)
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
            with torch.cuda.stream(side) if side is not None else _nullcontext():
                if args.gather == "full":
                    sharding.pack_solution(o["X"], o["U"], o["s"], out=packed[b])
                else:
                    packed[b].copy_(o["U"][:, 0, :])
                _, pending[b] = sharding.allgather_solutions(packed[b], Bg, dist, gathered=gathered[b], async_op=True)

        def drain():
            for b in range(len(pending)):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None

        for i in range(max(warmup, NH if budget > 0 else 0)):
            o_, side_ = solve_step(i)
            if world > 1:
                gather(i, o_, side_)
        if world > 1:
            drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            evs[i][0].record()
            o_, side_ = solve_step(i)
            evs[i][1].record()
            if world > 1:
                gather(i, o_, side_)
        if world > 1:
            drain()
        torch.cuda.synchronize()
        out = o_                                               # (every output set holds the same batch's solution)
        t_rank = time.perf_counter() - t0                      # this rank's own time for its K steps (before the barrier)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        # the gathered table of the last step against this rank's own solution (outside the timed region): every rank's slice of
        # the table must be its packed records, bit for bit
        gather_ok = None
        if world > 1:
            bl = (steps - 1) % len(pending)
            mine = sharding.pack_solution(out["X"], out["U"], out["s"]) if args.gather == "full" else out["U"][:, 0, :]
            ok = torch.equal(gathered[bl][lo:hi] if scaling == "strong" else gathered[bl][rank * Bl:(rank + 1) * Bl], mine) and bool(torch.isfinite(gathered[bl]).all())
            flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            gather_ok = bool(flag.item() == 1.0)
        step_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)     # solve kernel per step, HIP events on the launch stream
        status = out["status"].cpu().numpy()
        iters = out["iters"].cpu().numpy()
        err = out["err"].cpu().numpy()
        stats = torch.tensor([float((status == 0).sum()), float(iters.sum()), float(iters.max()) if Bl else 0.0, float(err.max()) if Bl else 0.0,
                              t_rank / steps * 1e3, step_ms[len(step_ms) // 2], float(Bl)], dtype=torch.float64, device=dev)
        if world > 1:
            parts = [torch.zeros_like(stats) for _ in range(world)]
            dist.all_gather(parts, stats)
            parts = [p.tolist() for p in parts]
        else:
            parts = [stats.tolist()]
        if budget > 0:
            for e_ in engs:
                e_.set_iteration_budget(0)
        return dict(Bg=Bg, Bl=Bl, el=el, steps=steps, value=Bg * steps / el, ms_per_step=el / steps * 1e3, step_ms=step_ms, parts=parts, out=out,
                    iters=iters, eng=eng, ctrl=ctrl, d=d, x_init=x_init, traj=traj, uref=uref, obs=obs, ulast=ulast, evs=evs, NH=NH,
                    gather_ok=gather_ok, keep=keep)

    # ---- the headline: one launch per batch on the plain kernel, at every N (the same mode on one GPU and on eight)
    budget = args.budget
    R = run_config(args.scaling, budget, args.steps, args.warmup)
    Bg, Bl, el, step_ms, parts, out, iters = R["Bg"], R["Bl"], R["el"], R["step_ms"], R["parts"], R["out"], R["iters"]
    eng, ctrl, d, x_init, traj, uref, obs, ulast, evs, NH = (R[k] for k in ("eng", "ctrl", "d", "x_init", "traj", "uref", "obs", "ulast", "evs", "NH"))
    n_conv = sum(p[0] for p in parts); it_sum = sum(p[1] for p in parts)
    it_max = max(p[2] for p in parts); err_max = max(p[3] for p in parts)
    # ---- collective extras (every rank takes part): the pipelined-budget mode on the same batches, and - in a weak-scaling
    #      multi-rank run - config C4 read literally (global batch 8192, 1024 per GPU at N = 8)
    extras = {}

    def pipelined_extra():
        P = run_config(args.scaling, args.pipelined_budget, args.steps, max(args.warmup, 2))
        r_ = {"iteration_budget": args.pipelined_budget, "handles": P["NH"], "value": P["value"], "unit": "solves/s",
              "ms_per_step": P["ms_per_step"], "converged_frac": sum(p[0] for p in P["parts"]) / P["Bg"],
              "gather_checked": P["gather_ok"],
              "per_rank_ms_per_step": [p[4] for p in P["parts"]],
              "note": "same batches, same results: at most %d iterations per launch, the instances that need more are "
                      "finished by continuation launches on side streams while the next batches run (%d handles in "
                      "rotation) - the throughput of a stream of batches does not wait for a batch's slowest instance; "
                      "no roofline figure for this mode (the continuation runs beside the next launch)" % (args.pipelined_budget, P["NH"])}
        del P
        return r_

    if not args.no_extras and world > 1:
        if budget == 0:
            extras["pipelined_budget"] = pipelined_extra()
        if world > 1 and args.scaling == "weak":
            S = run_config("strong", 0, args.steps, max(args.warmup, 2))
            extras["strong_scaling"] = {"value": S["value"], "unit": "solves/s", "ms_per_step": S["ms_per_step"], "global_batch": S["Bg"],
                                        "batch_per_gpu": [int(p[6]) for p in S["parts"]], "max_iters_per_rank": [int(p[2]) for p in S["parts"]],
                                        "converged_frac": sum(p[0] for p in S["parts"]) / S["Bg"], "gather_checked": S["gather_ok"],
                                        "note": "BASELINE config C4 read literally: ONE seeded batch of %d instances sharded over the ranks "
                                                "(contiguous slices) + the all-gather; a rank's time is its slowest instance" % S["Bg"]}
            del S

    if rank == 0:
        ms_per_step = el / args.steps * 1e3
        value = Bg * args.steps / el
        mean_iters = it_sum / Bg
        k_ms = sum(step_ms) / len(step_ms)                 # this rank's solve kernel, average launch duration
        # (with --budget > 0 the timed launches execute at most `budget` iterations per instance; the rest runs in the
        #  continuation launches beside the next step and is not part of this kernel's duration)
        it_kernel = float(np.minimum(iters, budget).sum()) if budget > 0 else float(iters.sum())
        fl = riccati_flops_per_iter(N, nx, nu, M, 4) * it_kernel
        achieved_tf = fl / (k_ms * 1e-3) / 1e12
        by = algorithmic_bytes_per_solve(N, nx, nu, M) * Bl
        traffic = None
        mfma = None
        tpath = os.path.join(_ROOT, "profiles", PROFILE_TAG + "_pmc_traffic.json")
        mpath = os.path.join(_ROOT, "profiles", PROFILE_TAG + "_pmc_mfma.json")
        if world == 1 and (N, M, Bg) == (20, 5, 8192) and os.path.exists(tpath):
            # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), collected
            # separately with the same command and committed under profiles/ (bench.py cannot run the profiler itself)
            traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
        if world == 1 and (N, M) == (20, 5) and os.path.exists(mpath):
            # matrix-core use of the same kernel from its PMC passes (tools/pmc_collect.sh): v_mfma_f64_16x16x4 count and
            # busy cycles per wave per solver iteration; utilisation = executed MFMA flops / s over the 78.6 TFLOP/s peak
            c = json.load(open(mpath))["per_wave_per_iteration"]
            n_mfma = c.get("SQ_INSTS_MFMA", 0.0)
            mfma = {"insts_per_iter": n_mfma, "busy_cycles_per_iter": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
                    "wave_cycles_per_iter": 4.0 * c.get("SQ_WAVE_CYCLES", 0.0),
                    "executed_tflops": n_mfma * 2048.0 * mean_iters * Bl / (k_ms * 1e-3) / 1e12,
                    "source": "profiles/%s_pmc_mfma.json" % PROFILE_TAG}
            mfma["util"] = mfma["executed_tflops"] / FP64_PEAK_TFLOPS
        res = {
            "metric": "MPC solves/sec, whole-body N=%d batch=%d" % (N, args.batch),
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "whole-body MPC solve, N=%d, M=%d static circle obstacles, batch %d per GPU%s (global %d), "
                                   "cold start (u_latest=0), seeded synthetic (x_init, traj_ref, obstacles), no schedule hint from "
                                   "earlier solves (launch order by the a-priori difficulty key of the batch's own data)"
                                   % (N, M, Bl, "" if args.scaling == "weak" else " [strong: global batch fixed]", Bg),
                       "batch_per_gpu": Bl,
                       "seeds": ([args.seed_base + r for r in range(world)] if args.scaling == "weak" else [args.seed_base]),
                       "parallelism": "batch-sharded x%d%s" % (world, (" + all-gather(%s)" % ("X,U,s" if args.gather == "full" else "u0")) if world > 1 else ""),
                       "iteration_budget": budget,
                       "iteration_budget_note": ("off" if budget == 0 else "at most %d iterations per launch; the instances that need more are "
                                                 "finished by a continuation launch on a side stream while the next batches run (%d handles in "
                                                 "rotation); same results as one uninterrupted solve" % (budget, NH))},
            "solver": {"converged_frac": n_conv / Bg, "mean_iters": mean_iters, "max_iters": it_max,
                       "max_scaled_kkt": err_max, "lds_bytes_per_problem": eng.lds_bytes, "problems_per_cu": eng.problems_per_cu},
            "median_kernel_ms": step_ms[len(step_ms) // 2],
            # what every rank did on its own (the job's time is the slowest rank's): a rank whose batch holds one instance of
            # several hundred iterations shows here as a long step with ordinary mean iterations
            "per_rank": [{"rank": r, "batch": int(p[6]), "ms_per_step": p[4], "median_kernel_ms": p[5], "max_iters": int(p[2]),
                          "mean_iters": p[1] / p[6], "converged": int(p[0])} for r, p in enumerate(parts)],
            "roofline": {"bound": "mfma", "kernel": "mmpc_fast_kernel<0,20,5>" if (N, M) == (20, 5) else "mmpc_solve_kernel<0>", "achieved": achieved_tf,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tf / FP64_PEAK_TFLOPS,
                         "traffic": traffic, "kernel_ms": k_ms, "mfma": mfma,
                         "flops_per_iter": riccati_flops_per_iter(N, nx, nu, M, 4),
                         # HBM side: from the counters' bytes when there are any (traffic), else the algorithmic bytes
                         "hbm_achieved_GBs": (traffic or by) / (k_ms * 1e-3) / 1e9, "hbm_frac": (traffic or by) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "hbm_bytes_source": "PMC counters (profiles/%s_pmc_traffic.json)" % PROFILE_TAG if traffic else "algorithmic bytes (no counter file for this shape)",
                         "algorithmic_bytes": by},
        }
        res.update(extras)
        if world > 1 and "strong_scaling" in extras:
            # BASELINE config C4 read literally (ONE batch of --batch instances sharded over the ranks) beside the weak-scaling
            # headline: `value` is the weak figure (--batch instances PER GPU), `value_strong` the literal one
            res["value_strong"] = extras["strong_scaling"]["value"]
            res["config"]["workload"] += "; `value` = weak scaling (%d instances per GPU), `value_strong` = config C4 literally (%d instances in all)" % (args.batch, args.batch)
        if world > 1:
            res["gather_checked"] = R["gather_ok"]        # every rank found its own records, bit for bit, in the gathered table
        ev0, ev1 = evs[0]
        if world == 1 and not args.no_cpu and not args.no_extras:   # (--no-cpu = the profiler passes: they see the warm-up and timed launches only)
            # Extras, outside the timed region.
            # (0) config C5 (N = 30, 8 moving obstacles, warm-started receding horizon) in short form: `python bench.py --config c5`
            # prints the full line.  FIRST among the extras: its groups run on their own HIP streams, and streams are mapped to the
            # runtime's hardware queues in the order they are created - after the streams of the other extras two groups shared a
            # queue (their kernels then run one after the other) and the compact figure read 383 k where the stand-alone process
            # reads 435 k (round 3)
            try:
                res["c5"] = run_c5(args, torch, mm, None, 0, 1, dev, local_dev, steps=2, warmup=4, cpu_instances=256, compact=True)
            except Exception as e:
                res["c5"] = {"error": repr(e)}
            # (1) the same launches in plain batch order (mode 0) and WITH the history hint
            # (mode 1, the engine's default): workgroups start longest-first by the iteration counts of the handle's previous
            # solve - exact here because the batch is re-solved, correlated in a receding-horizon loop.
            def timed_mode(mode):
                eng.set_schedule_hint(mode)
                ts = []
                for _ in range(6):
                    ev0.record()
                    eng.solve_batch_device(x_init, traj, uref, ulast, obs, out=out)
                    ev1.record(); ev1.synchronize()
                    ts.append(ev0.elapsed_time(ev1))
                return sorted(ts[1:])[2]
            b_ms, h_ms = timed_mode(0), timed_mode(1)
            res["schedule_hint"] = {"in_timed_steps": "a-priori difficulty key of the batch's own data (mode 2)",
                                    "batch_order_ms": b_ms, "batch_order_value": Bl / (b_ms * 1e-3),
                                    "hinted_ms": h_ms, "hinted_value": Bl / (h_ms * 1e-3), "unit": "solves/s",
                                    "note": "batch_order: mode 0; hinted: same batch re-solved with the longest-first order of its "
                                            "previous solve (exact hint) - not the headline; every instance converges to the same "
                                            "result in every order"}
            # (1b) iteration budget + continuation (mmpc_set_iteration_budget / mmpc_resume_batch_device): the main launch gives
            # every instance at most 32 iterations - the results of the ~97 % that converge by then are complete when it ends -,
            # the continuation launch finishes the suspended ones (bitwise the same results as one uninterrupted launch)
            eng.set_schedule_hint(2)
            eng.set_iteration_budget(32)
            cm, cr = [], []
            ev2 = torch.cuda.Event(enable_timing=True)
            for _ in range(4):
                ev0.record()
                eng.solve_batch_device(x_init, traj, uref, ulast, obs, out=out)
                ev1.record()
                eng.resume_batch_device(x_init, traj, uref, ulast, obs, out=out)
                ev2.record(); ev2.synchronize()
                cm.append(ev0.elapsed_time(ev1)); cr.append(ev1.elapsed_time(ev2))
            res["continuation"] = {"budget": 32, "main_launch_ms": sorted(cm)[1], "continuation_ms": sorted(cr)[1],
                                   "suspended_after_main": eng.suspended_count(),
                                   "note": "main launch: every instance gets at most 32 iterations; continuation: the suspended "
                                           "instances run to convergence; outputs bitwise equal to the single launch"}
            eng.set_iteration_budget(0)
            # (1c) the generic kernel (any N, M, dense weights, half-space rows, terminal equality - what C1 and the closed-loop
            # demo's 'approach' / 'manipulate' phases run on) on the same batch, for scale
            os.environ["MMPC_FORCE_GENERIC"] = "1"
            try:
                ctrlg = mm.MPCWholeBody(robot, [], [], N=N, max_batch=Bl, device=local_dev, n_obstacles=M)
            finally:
                del os.environ["MMPC_FORCE_GENERIC"]
            ctrlg._engine.set_schedule_hint(2)
            outg = ctrlg._engine.solve_batch_device(x_init, traj, uref, ulast, obs)
            ev0.record()
            outg = ctrlg._engine.solve_batch_device(x_init, traj, uref, ulast, obs, out=outg)
            ev1.record(); ev1.synchronize()
            g_ms = ev0.elapsed_time(ev1)
            res["generic_kernel"] = {"ms": g_ms, "value": Bl / (g_ms * 1e-3), "unit": "solves/s",
                                     "lds_bytes_per_problem": ctrlg._engine.lds_bytes, "problems_per_cu": ctrlg._engine.problems_per_cu,
                                     "max_abs_dX_vs_specialised": float((outg["X"] - out["X"]).abs().max()),
                                     "note": "mmpc_solve_kernel<0> (all state in LDS, Riccati pass on MFMA tiles) on the same batch"}
            del ctrlg, outg
            # (1d) the demo's shape (config C1: N = 20, three circle obstacles, two half-space planes): generic kernel with a
            # static LDS block (MMPC_STATIC_LIST), 2048 starts around the planes' ridge, intended rows and rows as written
            try:
                res["c1_shape_generic_kernel"] = c1_shape_extra(mm, robot, dev)
            except Exception as e:                                             # an extra: never takes the bench line down
                res["c1_shape_generic_kernel"] = {"error": repr(e)}
            # (2) two batches in flight (two handles, two HIP streams, alternating): the drain of one launch - waves still
            # iterating on its slowest instances while CUs idle - is filled by the next launch.  No hint.
            eng.set_schedule_hint(2)
            ctrl2 = mm.MPCWholeBody(robot, [], [], N=N, max_batch=Bl, device=local_dev, n_obstacles=M)
            ctrl2._engine.set_schedule_hint(2)
            engs, outs = (eng, ctrl2._engine), [out, None]
            streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            for rep in range(2 + args.steps):
                if rep == 2:
                    torch.cuda.synchronize(); p0 = time.perf_counter()
                for q in range(2):
                    with torch.cuda.stream(streams[q]):
                        outs[q] = engs[q].solve_batch_device(x_init, traj, uref, ulast, obs, out=outs[q])
            torch.cuda.synchronize()
            res["two_streams"] = {"value": 2 * args.steps * Bl / (time.perf_counter() - p0), "unit": "solves/s",
                                  "note": "two handles on two HIP streams, %d launches each, a-priori order; not the headline figure" % args.steps}
            # (3) PCIe-inclusive rate of the host-pointer entry point (mmpc_solve_batch: H2D, solve, D2H of X,U,s,...): a note
            # beside `value`, which is always the device-resident rate
            hx = np.clip(d["x_init"][:Bl], ctrl.xlim[0], ctrl.xlim[1])
            eng.reset(); eng.solve_batch(hx, d["traj_ref"][:Bl], d["u_ref"][:Bl], d["obs"][:Bl])
            h0 = time.perf_counter()
            for _ in range(3):
                eng.reset()
                eng.solve_batch(hx, d["traj_ref"][:Bl], d["u_ref"][:Bl], d["obs"][:Bl])
            res["host_pointer_api"] = {"value": 3 * Bl / (time.perf_counter() - h0), "unit": "solves/s",
                                       "note": "mmpc_solve_batch with pageable host arrays in and out (PCIe-inclusive, cold start)"}
            # (4) latency of ONE solve through the reference's own call (controller.solve(x_init, traj_ref, u_ref) -> u0), the
            # number the closed-loop driver sees per tick (interface_wholebody_qref.py:134); cold start each time
            import contextlib, io
            one = mm.MPCWholeBody(robot, [mm.Obstacles(*d["obs"][0, m]) for m in range(M)], [], N=N)
            lat = []
            with contextlib.redirect_stdout(io.StringIO()):
                for b in range(12):
                    one.reset()
                    l0 = time.perf_counter()
                    one.solve(d["x_init"][b].copy(), d["traj_ref"][b], d["u_ref"][b])
                    lat.append(time.perf_counter() - l0)
            res["single_solve_latency_ms"] = {"median": 1e3 * float(np.median(lat[2:])), "max": 1e3 * float(np.max(lat[2:])),
                                              "note": "MPCWholeBody.solve() for one instance, host arrays in, u0 out (10 solves)"}
            # (5) a four times larger resident batch (seeds s, s+1, s+2, s+3 of the generator in one launch): the drain of a launch
            # - CUs idling while the last waves finish - is a fixed cost, so bigger shards amortise it
            if (N, M) == (20, 5) and Bl == 8192:
                parts = [d] + [synth.make_batch(Bl, N=N, M=M, config_id=args.seed_base + q) for q in (1, 2, 3)]
                cat = lambda key: torch.from_numpy(np.ascontiguousarray(np.concatenate([p_[key] for p_ in parts]))).to(dev)
                ctrl4 = mm.MPCWholeBody(robot, [], [], N=N, max_batch=4 * Bl, device=local_dev, n_obstacles=M)
                ctrl4._engine.set_schedule_hint(2)
                x4 = torch.from_numpy(np.clip(np.concatenate([p_["x_init"] for p_ in parts]), ctrl.xlim[0], ctrl.xlim[1])).to(dev)
                t4, u4, o4 = cat("traj_ref"), cat("u_ref"), cat("obs")
                ul4 = torch.zeros((4 * Bl, N, nu), dtype=torch.float64, device=dev)
                out4 = ctrl4._engine.solve_batch_device(x4, t4, u4, ul4, o4)
                ts = []
                for _ in range(3):
                    ev0.record()
                    ctrl4._engine.solve_batch_device(x4, t4, u4, ul4, o4, out=out4)
                    ev1.record(); ev1.synchronize()
                    ts.append(ev0.elapsed_time(ev1))
                l_ms = sorted(ts)[1]
                res["large_batch"] = {"batch": 4 * Bl, "ms": l_ms, "value": 4 * Bl / (l_ms * 1e-3), "unit": "solves/s",
                                      "max_iters": int(out4["iters"].max()), "converged_frac": float((out4["status"] == 0).double().mean()),
                                      "note": "one launch of 32768 instances (seeds %d..%d), a-priori order" % (args.seed_base, args.seed_base + 3)}
                del ctrl4, out4, x4, t4, u4, o4, ul4
                # (6) a stream of DIFFERENT batches, some of which hold a straggler (seeds s..s+7: what the ranks of an 8-GPU run
                # hold; slowest instances 84, 66, 72, 146, 238, 456, 88, 68 iterations): one after the other, and with an
                # iteration budget of 64 + continuation launches on side streams while the next batches run
                try:
                    res["stream_of_batches"] = stream_of_batches_extra(mm, robot, dev, local_dev, args.seed_base, N, M, Bl, nu, ctrl.xlim)
                except Exception as e:
                    res["stream_of_batches"] = {"error": repr(e)}
            # (6b) the pipelined-budget mode on the headline's own batch (the collective extra of the multi-rank runs, here at N = 1;
            # after the other extras: its side streams would otherwise take the hardware queues of theirs)
            if budget == 0:
                try:
                    res["pipelined_budget"] = pipelined_extra()
                except Exception as e:
                    res["pipelined_budget"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(d, N, M, min(args.cpu_sample, Bl), out["X"])
        # one word per extra and the figures the documents quote, flat, where a reader of the line's tail finds them
        names = ("c5", "schedule_hint", "continuation", "generic_kernel", "c1_shape_generic_kernel", "two_streams", "host_pointer_api",
                 "single_solve_latency_ms", "large_batch", "stream_of_batches", "pipelined_budget", "strong_scaling", "cpu_baseline")
        res["extras_status"] = {k: ("error" if isinstance(res[k], dict) and "error" in res[k] else "ok") for k in names if k in res}
        g = lambda *path: _dig(res, path)
        res["extras_summary"] = {k: v for k, v in {
            "generic_kernel_c4_solves_s": g("generic_kernel", "value"),
            "c1_as_written_solves_s": g("c1_shape_generic_kernel", "rows_as_written", "value"),
            "c1_as_written_converged_frac": g("c1_shape_generic_kernel", "rows_as_written", "converged_frac"),
            "c5_lock_step_solves_s": g("c5", "value"),
            "c5_best_groups_solves_s": g("c5", "groups_out_of_phase", "best", "value"),
            "c5_converged_frac": g("c5", "converged_frac"),
            "hinted_ms": g("schedule_hint", "hinted_ms"),
            "batch_order_ms": g("schedule_hint", "batch_order_ms"),
            "two_streams_solves_s": g("two_streams", "value"),
            "stream_of_batches_plain_ms": g("stream_of_batches", "plain", "ms_per_batch"),
            "single_solve_latency_ms": g("single_solve_latency_ms", "median"),
            "cpu_port_solves_s": g("cpu_baseline", "value"),
        }.items() if v is not None}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def stream_of_batches_extra(mm, robot, dev, local_dev, seed0, N, M, B, nu, xlim, nseeds=8, rounds=5, budget=64, nh=8, nside=6):
    """Eight different seeded batches solved in a stream, `rounds` times over: plain (one launch per batch, a-priori order) and
    with the iteration budget + pipelined continuation (as the multi-rank runs use it).  Every output is checked for status 0."""
    import torch
    synth = _synth()
    data = []
    for q in range(nseeds):
        d = synth.make_batch(B, N=N, M=M, config_id=seed0 + q)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        data.append((t(np.clip(d["x_init"], xlim[0], xlim[1])), t(d["traj_ref"]), t(d["u_ref"]), t(d["obs"])))
    ul = torch.zeros((B, N, nu), dtype=torch.float64, device=dev)
    ctrls = [mm.MPCWholeBody(robot, [], [], N=N, max_batch=B, device=local_dev, n_obstacles=M) for _ in range(nh)]
    engs = [c._engine for c in ctrls]
    for e in engs:
        e.set_schedule_hint(2)
    outs = [None] * nh
    sides = [torch.cuda.Stream(device=dev) for _ in range(nside)]
    res = {"seeds": [seed0 + q for q in range(nseeds)], "batches": nseeds * rounds, "unit": "ms per batch", "iteration_budget": budget,
           "note": "plain: one launch per batch; pipelined: at most %d iterations per launch, the rest by continuation launches on side "
                   "streams while the next batches run (%d handles, %d side streams); same results" % (budget, nh, nside)}
    for mode in ("plain", "pipelined"):
        for e in engs:
            e.set_iteration_budget(budget if mode == "pipelined" else 0)
        for rep in range(2):                                  # first pass: warm-up
            torch.cuda.synchronize(); t0 = time.perf_counter()
            bad = 0
            for i in range(nseeds * rounds):
                x, tr, ur, ob = data[i % nseeds]
                h = i % nh
                if mode == "plain":
                    outs[h] = engs[h].solve_batch_device(x, tr, ur, ul, ob, out=outs[h])
                else:
                    outs[h] = engs[h].solve_batch_device(x, tr, ur, ul, ob, out=outs[h])
                    engs[h].resume_batch_device(x, tr, ur, ul, ob, outs[h], stream=sides[i % nside].cuda_stream)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        conv = min(float((o["status"] == 0).double().mean()) for o in outs if o is not None)
        res[mode] = {"ms_per_batch": 1e3 * el / (nseeds * rounds), "value": B * nseeds * rounds / el, "converged_frac_min": conv}
    for e in engs:
        e.set_iteration_budget(0)
    return res


def c1_shape_extra(mm, robot, dev, B=2048, N=20):
    """Generic kernel on the demo's shape (starts as in tests/test_gpu_certificates.py::test_as_written_halfspace_rows_batch)."""
    import torch
    synth = _synth()
    x, tr, obs, hs = synth.make_c1_starts(B, N)
    oml = [(h[:3], h[3:].reshape(1, 3)) for h in hs]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xi, trd, ob = t(x), t(tr), t(obs)
    z = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
    out = {"unit": "solves/s", "batch": B, "note": "mmpc_solve_kernel_static<0,20,3,0,2,*>: all state in LDS, Riccati pass on MFMA tiles; a-priori order"}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, fc in (("intended_rows", False), ("rows_as_written", None)):
        ctrl = mm.MPCWholeBody(robot, [], oml, N=N, max_batch=B, device=dev.index if dev.index is not None else 0, n_obstacles=3, faithful_convex=fc)
        eng = ctrl._engine
        eng.set_schedule_hint(2)
        o = eng.solve_batch_device(xi, trd, z, z, ob)
        e0.record(); o = eng.solve_batch_device(xi, trd, z, z, ob, out=o); e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1)
        out[name] = {"ms": ms, "value": B / (ms * 1e-3), "converged_frac": float((o["status"] == 0).double().mean()),
                     "mean_iters": float(o["iters"].double().mean()), "max_iters": int(o["iters"].max()),
                     "lds_bytes_per_problem": eng.lds_bytes, "problems_per_cu": eng.problems_per_cu}
        if fc is None:
            # the CPU port on the same shape (NLP as written), and the latency of ONE solve through the reference's own call
            from oracle import coracle, nlp
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = min(avail, 16)
            par = nlp.WholeBodyParams(N=N)
            zc = np.zeros((B, N, 5))
            c0 = time.perf_counter()
            oc = coracle.solve_batch(par, x, tr, zc, zc, obs, hs=hs, as_written=True, nthreads=cores, max_iter=2000)
            ct = time.perf_counter() - c0
            gX = o["X"].cpu().numpy()
            same = np.abs(oc["cost"] / o["cost"].cpu().numpy() - 1) < 1e-6
            out["cpu_baseline"] = {"value": B / ct, "unit": "solves/s", "cores": cores, "kind": "port",
                                   "sample": "the same %d starts, NLP as written, oracle/mmpc_oracle.c (OpenMP), one pass, %.1f s" % (B, ct),
                                   "converged_frac": float((oc["status"] == 0).mean()), "same_minimum_as_gpu": int(same.sum()),
                                   "max_abs_dX_vs_gpu_same_minimum": float(np.abs(gX[same] - oc["X"][same]).max())}
            import contextlib, io
            obst = [mm.Obstacles(*obs[0, m]) for m in range(3)]
            one = mm.MPCWholeBody(robot, obst, oml, N=N)
            lat = []
            with contextlib.redirect_stdout(io.StringIO()):
                for b in range(12):
                    one.reset()
                    l0 = time.perf_counter()
                    one.solve(x[b].copy(), tr[b], np.zeros((N, 5)))
                    lat.append(time.perf_counter() - l0)
            out["single_solve_latency_ms"] = {"median": 1e3 * float(np.median(lat[2:])), "max": 1e3 * float(np.max(lat[2:])),
                                              "note": "MPCWholeBody.solve() of the demo's controller (two planes, NLP as written) for one start: "
                                                      "host arrays in, u0 out (10 solves, generic kernel)"}
        del ctrl, o
    return out


def cpu_baseline(d, N, M, ns, gpu_X, obs=None, u_last=None, x_init=None, traj=None, seconds=10.0):
    """The CPU restatement (oracle/mmpc_oracle.c, OpenMP over the batch) timed on the GPU box's host cores on a bounded
    sample of the same workload; `kind` is "port": the reference's own solver (CasADi/IPOPT) cannot be installed here."""
    from oracle import coracle, nlp
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, 16)          # the GPU box gives one GPU a 16-core CPU share
    par = nlp.WholeBodyParams(N=N)
    xi = np.clip(d["x_init"][:ns], par.xlim[0], par.xlim[1]) if x_init is None else x_init[:ns]
    tr = d["traj_ref"][:ns] if traj is None else traj[:ns]
    ob = d["obs"][:ns] if obs is None else obs[:ns]
    ul = np.zeros((ns, N, 5)) if u_last is None else u_last[:ns]
    coracle.lib()
    # bounded sample: passes over the sample until about 10 s of wall time on the host cores (at most 24 passes)
    c0 = time.perf_counter(); passes = 0
    while passes < 24 and (passes == 0 or time.perf_counter() - c0 < seconds):
        o = coracle.solve_batch(par, xi, tr, d["u_ref"][:ns], ul, ob, nthreads=cores, max_iter=2000)
        passes += 1
    ct = (time.perf_counter() - c0) / passes
    n1 = min(512, ns)                                     # the same port on ONE core (a 512-instance sample)
    c1 = time.perf_counter()
    coracle.solve_batch(par, xi[:n1], tr[:n1], d["u_ref"][:n1], ul[:n1], ob[:n1], nthreads=1, max_iter=2000)
    one_core = n1 / (time.perf_counter() - c1)
    gX = gpu_X[:ns].cpu().numpy()
    dev = np.abs(gX - o["X"]).reshape(ns, -1).max(1)
    return {"value": ns / ct, "unit": "solves/s", "cores": cores, "kind": "port", "value_1_core": one_core,
            "sample": "first %d instances of the same batch, oracle/mmpc_oracle.c (OpenMP), %d passes, %.1f s in all" % (ns, passes, ct * passes),
            "max_abs_dX_vs_gpu": float(dev.max()), "n_dX_above_1e-6": int((dev > 1e-6).sum()),
            "casadi": "CasADi/IPOPT baseline unavailable on this host" if not _has_casadi() else "importable"}


def main_c5(args):
    """`--config c5`: BASELINE.json configs[4] as its own bench line, on one GPU or - under torchrun - on N (one rank per GPU)."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import mmpc_loader
    mm = mmpc_loader.load()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    backend = os.environ.get("MMPC_BENCH_BACKEND", "nccl")
    local_dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    res = run_c5(args, torch, mm, dist, rank, world, dev, local_dev, steps=args.steps, warmup=args.warmup,
                 cpu_instances=0 if (args.no_cpu or world > 1) else 1024, compact=False)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def run_c5(args, torch, mm, dist, rank, world, dev, local_dev, steps, warmup, cpu_instances, compact):
    """Config C5 (BASELINE.json configs[4]): N=30, M=8 moving obstacles (centre c + v (tick+k) dt at stage k: the build's
    definition, no reference code exists), `--ticks` receding-horizon ticks, tick 0 cold, later ticks warm-started per
    mpc_wholebody_qref.py:303,310 (U init = U_last = previous optimum, X init = tile(x_init)).  Plant step and local-reference
    window (interface_wholebody_qref.py:143,353-396) run as torch ops on the device.  One 'step' = all ticks;
    value = robots x ticks / time.
    N ranks: the robots are sharded like the instances of C4 - weak (default): every rank simulates its own `--batch` robots
    (seed 5 + rank), strong: one population of `--batch` robots in contiguous slices -, every rank keeps its slice of u_latest
    and of the robots' states resident, and the one exchange per tick is the all-gather of the first inputs u0 the closed loop
    applies (5 doubles per robot), asynchronous: it travels while the next tick is solved."""
    synth = _synth()
    from mmpc_amd import sharding
    N, M, T = 30, 8, args.ticks
    Bg = args.batch * (world if args.scaling == "weak" else 1)
    lo, hi = sharding.shard_bounds(Bg, world, rank)
    B = hi - lo
    if args.scaling == "weak":
        d = synth.make_batch(args.batch, N=N, M=M, config_id=5 + rank, moving=True)
        sl = slice(0, B)
    else:
        d = synth.make_batch(Bg, N=N, M=M, config_id=5, moving=True)
        sl = slice(lo, hi)
    ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=max(B, 1), device=local_dev, n_obstacles=M, obs_per_stage=True)
    eng = ctrl._engine
    f64 = dict(dtype=torch.float64, device=dev)
    x0 = torch.from_numpy(np.clip(d["x_init"][sl], ctrl.xlim[0], ctrl.xlim[1])).to(dev)
    glob = torch.from_numpy(np.ascontiguousarray(d["traj_ref"][sl])).to(dev)     # first N+1=31 rows of the 51-row global plan
    step = (glob[:, N] - glob[:, 0]) / N
    glob = glob[:, :1] + step[:, None, :] * torch.arange(51, **f64)[None, :, None]   # full 51-row straight-line plan
    obs0 = torch.from_numpy(np.ascontiguousarray(d["obs"][sl])).to(dev); vel = torch.from_numpy(np.ascontiguousarray(d["obs_vel"][sl])).to(dev)
    uref = torch.zeros((B, N, 5), **f64)
    xlo = torch.from_numpy(ctrl.xlim[0]).to(dev); xhi = torch.from_numpy(ctrl.xlim[1]).to(dev)
    karr = torch.arange(N + 1, **f64)

    def f_batch(xc, u):
        c, s = torch.cos(xc[:, 2]), torch.sin(xc[:, 2])
        return torch.stack([xc[:, 0] + 0.1 * xc[:, 3], xc[:, 1] + 0.1 * xc[:, 4], xc[:, 2] + 0.1 * xc[:, 5],
                            xc[:, 3] + 0.1 * (u[:, 0] * c - xc[:, 4] * xc[:, 5]),
                            xc[:, 4] + 0.1 * (u[:, 0] * s + xc[:, 3] * xc[:, 5]), xc[:, 5] + 0.1 * u[:, 1],
                            xc[:, 6] + 0.1 * u[:, 2], xc[:, 7] + 0.1 * u[:, 3], xc[:, 8] + 0.1 * u[:, 4]], dim=1)

    ug_buf = torch.zeros((B, N, 5), **f64); xg_buf = torch.zeros((B, N + 1, 9), **f64)
    evp = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(T)]
    cap = {}
    # u0 of every robot of the job, gathered per tick (two buffer sets: the gather of tick t travels while tick t+1 is solved)
    u0_loc = [torch.empty((B, 5), **f64) for _ in range(2)]
    u0_all = [torch.empty((Bg, 5), **f64) for _ in range(2)]
    pend = [None, None]
    gather_ok = [True]

    def run_all(shifted=False, timed=False, check=False):
        """shifted=False: the reference's protocol (U starts at the previous optimum unshifted = U_last, X at tile(x_init),
        cold barrier parameter).  shifted=True: the engine's opt-in warm start (mmpc_set_warm_start) from tick 1 on - U starts
        at the previous optimum shifted by one stage, X at its roll-out, mu at 0.1; U_last, and with it the NLP, is unchanged."""
        x = x0.clone(); ul = torch.zeros((B, N, 5), **f64); its = []
        out = None
        eng.set_warm_start(None, 1.0)
        eng.reset()      # forget the schedule hint of the previous pass: tick 0 is a first solve, later ticks are hinted by the tick before
        for t in range(T):
            dist_ = torch.linalg.norm(x[:, None, :2] - glob[:, :, :2], dim=2)
            start = torch.argmin(dist_, dim=1)
            idx = torch.clamp(start[:, None] + torch.arange(N + 1, device=dev)[None, :], max=50)
            loc = torch.gather(glob, 1, idx[:, :, None].expand(B, N + 1, 9)).contiguous()
            obs = obs0[:, None, :, :].repeat(1, N + 1, 1, 1)
            obs[..., :2] += vel[:, None, :, :] * ((t + karr) * 0.1)[None, :, None, None]
            xg = None
            if shifted and t >= 1:
                ug_buf[:, :-1] = ul[:, 1:]; ug_buf[:, -1] = ul[:, -1]
                xg_buf[:, 0] = torch.minimum(torch.maximum(x, xlo), xhi)
                for k in range(N):
                    xg_buf[:, k + 1] = f_batch(xg_buf[:, k], ug_buf[:, k])
                if t == 1:
                    eng.set_warm_start(ug_buf, 0.1)
                xg = xg_buf
            obs = obs.contiguous()
            if timed:
                evp[t][0].record()
            out = eng.solve_batch_device(x, loc, uref, ul, obs, x_guess=xg, out=out)
            if timed:
                evp[t][1].record()
            if timed and t in (0, T - 1) and "x%d" % t not in cap:      # inputs/outputs of two ticks for the CPU leg's parity figure
                torch.cuda.synchronize()
                cap["x%d" % t] = x.clone(); cap["loc%d" % t] = loc.clone(); cap["ul%d" % t] = ul.clone(); cap["obs%d" % t] = obs.clone()
                cap["X%d" % t] = out["X"].clone()
            ul = out["U"].clone()
            u0 = out["U"][:, 0]
            if world > 1:
                b = t & 1
                if pend[b] is not None:
                    pend[b].wait(); pend[b] = None
                u0_loc[b].copy_(u0)
                _, pend[b] = sharding.allgather_solutions(u0_loc[b], Bg, dist, gathered=u0_all[b], async_op=True)
                if check and t == T - 1:
                    pend[b].wait(); pend[b] = None
                    gather_ok[0] = gather_ok[0] and torch.equal(u0_all[b][lo:hi], u0_loc[b]) and bool(torch.isfinite(u0_all[b]).all())
            x = f_batch(torch.minimum(torch.maximum(x, xlo), xhi), u0)
            its.append((out["iters"].double().mean() if B else torch.zeros((), **f64), (out["status"] == 0).double().sum(), out["iters"].max() if B else torch.zeros((), **f64)))
        for b in range(2):
            if pend[b] is not None:
                pend[b].wait(); pend[b] = None
        return its

    def timed_passes(shifted):
        for _ in range(max(1, warmup // 2)):
            run_all(shifted=shifted)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            its = run_all(shifted=shifted)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], **f64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, its

    el, its = timed_passes(False)
    # kernel time and iteration totals of one more pass (HIP events around every tick's solve launch, on its stream)
    its_k = run_all(timed=True, check=True)
    torch.cuda.synchronize()
    k_ms = [e0.elapsed_time(e1) for e0, e1 in evp]
    tot_iters = sum(float(a) for a, _, _ in its_k) * B
    fl_iter = riccati_flops_per_iter(N, 9, 5, M, 4)
    ach = fl_iter * tot_iters / (sum(k_ms) * 1e-3) / 1e12
    by = 8 * (9 + 9 * (N + 1) + 5 * N + 5 * N + 3 * M * (N + 1)) + 8 * (9 * (N + 1) + 5 * N + (N + 1))   # per solve, per-stage obstacle table
    # the opt-in warm start, timed the same way (reported beside the figure of the reference's protocol)
    el_w, its_w = timed_passes(True)
    eng.set_warm_start(None, 1.0)
    # per-rank summary (converged solves, slowest instance per tick)
    conv_loc = torch.tensor([sum(float(b_) for _, b_, _ in its), float(B * T), max(float(c_) for _, _, c_ in its), sum(k_ms)], **f64)
    if world > 1:
        allp = [torch.zeros_like(conv_loc) for _ in range(world)]
        dist.all_gather(allp, conv_loc)
        allp = [p.tolist() for p in allp]
        flag = torch.tensor([1.0 if gather_ok[0] else 0.0], **f64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gok = bool(flag.item() == 1.0)
    else:
        allp = [conv_loc.tolist()]
        gok = None
    c5_tr = _c5_traffic() if (world == 1 and B == 8192) else None
    res = {"metric": "MPC solves/sec, whole-body N=30 batch=%d, %d warm-started receding-horizon ticks, 8 moving obstacles" % (args.batch, T),
           "value": Bg * T * steps / el, "unit": "solves/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "C5: whole-body N=30, M=8 moving circle obstacles, %d robots per GPU (global %d), %d ticks (tick 0 cold; the "
                                  "launch order of a tick is hinted by the iteration counts of the tick before)" % (B, Bg, T),
                      "batch_per_gpu": B, "parallelism": "robots sharded x%d%s" % (world, " + all-gather(u0) per tick" if world > 1 else "")},
           "roofline": {"bound": "mfma", "kernel": "mmpc_fast_kernel<0,30,8>", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / FP64_PEAK_TFLOPS, "traffic": c5_tr, "kernel_ms_per_tick": k_ms, "flops_per_iter": fl_iter,
                        # HBM side from the counters' bytes per launch (= per tick) when there are any: the gain block of the long
                        # horizons is written twice per iteration through L2, 47x the algorithmic bytes
                        "algorithmic_bytes_per_tick": by * B,
                        "traffic_over_algorithmic": (c5_tr / (by * B)) if c5_tr else None,
                        "hbm_achieved_GBs": (c5_tr or by * B) * T / (sum(k_ms) * 1e-3) / 1e9,
                        "hbm_frac": (c5_tr or by * B) * T / (sum(k_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "hbm_bytes_source": "PMC counters (profiles/%s_c5_pmc_traffic.json)" % PROFILE_TAG if c5_tr else "algorithmic bytes"},
           "solver": {"mean_iters_per_tick": [float(a) for a, _, _ in its], "converged_frac": sum(p[0] for p in allp) / max(sum(p[1] for p in allp), 1.0),
                      "max_iters_per_tick": [int(c) for _, _, c in its],
                      "lds_bytes_per_problem": eng.lds_bytes, "problems_per_cu": eng.problems_per_cu},
           "shifted_warm_start": {"value": Bg * T * steps / el_w, "unit": "solves/s", "ms_per_step": el_w / steps * 1e3,
                                  "mean_iters_per_tick": [float(a) for a, _, _ in its_w],
                                  "max_iters_per_tick": [int(c) for _, _, c in its_w],
                                  "note": "opt-in mmpc_set_warm_start: U guess = previous optimum shifted one stage, X guess = its "
                                          "roll-out (torch ops, inside the timed region), mu_init 0.1; same NLP, not the reference's protocol"}}
    if world > 1:
        res["per_rank"] = [{"rank": r, "converged": int(p[0]), "solves": int(p[1]), "max_iters": int(p[2]), "kernel_ms_per_pass": p[3]} for r, p in enumerate(allp)]
        res["gather_checked"] = gok
        if not args.no_extras and B >= 96:
            # collective extra: every rank's robots as G groups in lock step on their own streams, out of phase
            # (DeviceFleet.run_groups), the u0 all-gather issued per group and tick on the group's stream - a tick's exchange then
            # waits for the slowest robot of one group of one rank, not of the whole node.  Same per-robot numbers as lock step.
            G = 3
            sizes_equal = len({sharding.shard_bounds(Bg, world, r)[1] - sharding.shard_bounds(Bg, world, r)[0] for r in range(world)}) == 1
            del ctrl, eng
            fleet = mm.DeviceFleet(mm, x0, glob, obs0, vel, N=N, device=local_dev)
            gb = {}
            okg = [True]
            def on_tick(g, glo, ghi, t, u0):
                k = (g, t & 1)
                if k not in gb:
                    gb[k] = [torch.empty((ghi - glo, 5), **f64), torch.empty(((ghi - glo) * world, 5), **f64), None]
                loc_, all_, w_ = gb[k]
                if w_ is not None:
                    w_.wait()
                loc_.copy_(u0)
                if sizes_equal:
                    gb[k][2] = dist.all_gather_into_tensor(all_, loc_, async_op=True)
                else:                                       # ragged shards: group sizes differ between ranks, padded blocking gather
                    sharding.allgather_solutions(loc_, None, dist)
                if t == T - 1 and sizes_equal:
                    gb[k][2].wait(); gb[k][2] = None
                    okg[0] = okg[0] and torch.equal(all_[rank * (ghi - glo):(rank + 1) * (ghi - glo)], loc_) and bool(torch.isfinite(all_).all())
            def drain():
                for v in gb.values():
                    if v[2] is not None:
                        v[2].wait(); v[2] = None
            ref_ls = fleet.run_lockstep(T); torch.cuda.synchronize()
            fleet.run_groups(T, groups=G, on_tick=on_tick if sizes_equal else None); drain(); torch.cuda.synchronize()
            dist.barrier(); torch.cuda.synchronize(); g0 = time.perf_counter()
            for _ in range(steps):
                rg = fleet.run_groups(T, groups=G, on_tick=on_tick if sizes_equal else None)
            drain(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            tg = torch.tensor([time.perf_counter() - g0], **f64); dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            flag = torch.tensor([1.0 if (okg[0] and bool(rg["all_converged"]) and torch.equal(rg["u0"], ref_ls["u0"])) else 0.0], **f64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            res["groups_per_rank"] = {"groups": G, "value": Bg * T * steps / float(tg.item()), "unit": "solves/s", "ms_per_step": float(tg.item()) / steps * 1e3,
                                      "gather": "u0 per group and tick, asynchronous, on the group's stream" if sizes_equal else "none (ragged shards)",
                                      "checked": bool(flag.item() == 1.0),
                                      "note": "every rank's robots in %d groups out of phase (own stream, handle, descending priority); per-robot results "
                                              "bitwise those of lock step, every solve converged and every rank's rows found in the gathered tables when `checked`" % G}
            del fleet
    elif not args.no_extras:
        # the same fleet driven asynchronously (mmpc_amd.fleet.DeviceFleet.run_async: iteration budget per launch, the robots that
        # converge move on, the suspended ones are continued on a side stream and rejoin later), against the same class's lock step
        try:
            del ctrl, eng
            fleet = mm.DeviceFleet(mm, x0, glob, obs0, vel, N=N, device=local_dev)
            fa = {}
            for mode in ("lockstep", "async", "groups2", "groups3", "groups4"):
                if mode.startswith("groups") and B < 1024:
                    continue
                fn = (lambda: fleet.run_lockstep(T)) if mode == "lockstep" else ((lambda: fleet.run_async(T, budget=args.async_budget)) if mode == "async" else
                                                                              (lambda: fleet.run_groups(T, groups=int(mode[6:]))))
                fn(); torch.cuda.synchronize()
                f0 = time.perf_counter()
                for _ in range(steps):
                    r_ = fn()
                torch.cuda.synchronize()
                fa[mode] = (time.perf_counter() - f0, r_)
            same = bool(torch.equal(fa["lockstep"][1]["u0"], fa["async"][1]["u0"]) and torch.equal(fa["lockstep"][1]["x"], fa["async"][1]["x"]))
            res["async_receding_horizon"] = {
                "value": B * T * steps / fa["async"][0], "unit": "solves/s", "ms_per_step": fa["async"][0] / steps * 1e3,
                "lockstep_value": B * T * steps / fa["lockstep"][0], "lockstep_ms_per_step": fa["lockstep"][0] / steps * 1e3,
                "iteration_budget": args.async_budget, "rounds": fa["async"][1]["rounds"], "suspended_solves": int(fa["async"][1]["suspended"]),
                "all_converged": bool(fa["async"][1]["all_converged"]), "bitwise_equal_to_lockstep": same,
                "note": "same robots, same per-robot results: a launch gives a robot at most %d iterations; robots that converge take "
                        "their plant step and are launched again in the next round (device-side list, mmpc_solve_list_device), suspended "
                        "ones are continued on a side stream and rejoin two rounds later; the reference's warm-start protocol" % args.async_budget}
            res["groups_out_of_phase"] = {
                "note": "same robots, same per-robot results (reference warm-start protocol): the fleet as G contiguous groups, each in lock step on its own "
                        "HIP stream and handle at descending stream priority, so that one group's tail overlaps another group's bulk "
                        "(mmpc_amd.fleet.DeviceFleet.run_groups)",
                "unit": "solves/s", "lockstep_value": B * T * steps / fa["lockstep"][0]}
            for mode in fa:
                if mode.startswith("groups"):
                    res["groups_out_of_phase"][mode[6:]] = {
                        "value": B * T * steps / fa[mode][0], "ms_per_step": fa[mode][0] / steps * 1e3, "all_converged": bool(fa[mode][1]["all_converged"]),
                        "bitwise_equal_to_lockstep": bool(torch.equal(fa["lockstep"][1]["u0"], fa[mode][1]["u0"]) and torch.equal(fa["lockstep"][1]["x"], fa[mode][1]["x"])
                                                          and torch.equal(fa["lockstep"][1]["iters"], fa[mode][1]["iters"]))}
            gv = {k: v["value"] for k, v in res["groups_out_of_phase"].items() if isinstance(v, dict)}
            if gv:
                best = max(gv, key=gv.get)
                res["groups_out_of_phase"]["best"] = {"groups": int(best), "value": gv[best]}
            del fleet, fa
        except Exception as e:
            import traceback
            res["async_receding_horizon"] = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
    if cpu_instances > 0 and rank == 0:
        # CPU leg: the same two captured ticks (the cold one and the last warm one), first instances, C oracle on the host cores
        ns = min(cpu_instances, B)
        legs = []
        for t in (0, T - 1):
            dd = {"x_init": cap["x%d" % t].cpu().numpy(), "traj_ref": cap["loc%d" % t].cpu().numpy(), "u_ref": np.zeros((B, N, 5)),
                  "obs": cap["obs%d" % t].cpu().numpy()}
            legs.append(cpu_baseline(dd, N, M, ns, cap["X%d" % t], u_last=cap["ul%d" % t].cpu().numpy(), seconds=4.0 if compact else 10.0))
        res["cpu_baseline"] = {"value": 2.0 / (1.0 / legs[0]["value"] + 1.0 / legs[1]["value"]), "unit": "solves/s", "cores": legs[0]["cores"],
                               "kind": "port", "sample": "ticks 0 and %d of the timed loop, first %d instances each: " % (T - 1, ns) + legs[0]["sample"],
                               "per_tick": legs, "casadi": legs[0]["casadi"]}
    if compact:
        keep = ("value", "unit", "ms_per_step", "steps", "roofline", "cpu_baseline", "async_receding_horizon", "groups_out_of_phase")
        out = {k: res[k] for k in keep if k in res}
        out["workload"] = res["config"]["workload"]
        out["max_iters_per_tick"] = res["solver"]["max_iters_per_tick"]; out["mean_iters_per_tick"] = res["solver"]["mean_iters_per_tick"]
        out["converged_frac"] = res["solver"]["converged_frac"]
        out["shifted_warm_start_value"] = res["shifted_warm_start"]["value"]
        if "cpu_baseline" in out:
            out["cpu_baseline"] = {k: v for k, v in out["cpu_baseline"].items() if k != "per_tick"}
        return out
    return res


def _c5_traffic():
    p = os.path.join(_ROOT, "profiles", PROFILE_TAG + "_c5_pmc_traffic.json")
    return json.load(open(p))["traffic_bytes_per_launch"] if os.path.exists(p) else None


def _has_casadi():
    try:
        import casadi  # noqa: F401
        return True
    except Exception:
        return False


if __name__ == "__main__":
    main()
