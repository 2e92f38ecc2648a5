"""Device-resident receding-horizon loop over a fleet of B robots with moving obstacles (BASELINE config C5).

The per-robot loop is the reference's (interface_wholebody_qref.py:100-143 with physical_sim=False): local reference
window from the robot's global plan (:353-396), controller.solve(x, traj_ref, u_ref) (:134), plant step x <- f(x, u0)
(:143), warm start per mpc_wholebody_qref.py:301-310 (U init = U_last = the robot's previous optimum, unshifted; X init =
tile(x_init)).  Obstacle j of robot b sits at c_j + v_j (tick + k) dt at stage k of tick `tick` (the build's definition of
the moving-obstacle scenario: README.md:57,85-88 only names the branch).  Everything stays in HBM between ticks; PyTorch
owns the memory and the streams, every solve goes through the C ABI.

Two drivers over the same robots, with bitwise equal per-robot results (tests/test_gpu_parity.py):
  * run_lockstep: all robots solve tick t together, one launch per tick - a tick lasts as long as its slowest robot
    (one wave solves one instance: 200-450 iterations for one robot in 8192 against a mean of 34);
  * run_async: every launch gives a robot at most `budget` iterations (mmpc_set_iteration_budget).  The robots that
    converge within it take their plant step and are launched again in the next round (mmpc_solve_list_device: a device-side
    list of the robots that are ready, longest previous solve first); the few that were suspended are continued on a side
    stream (mmpc_resume_batch_device) while the others move on, and rejoin two rounds later, one tick behind.  Two handles /
    buffer sets alternate, so that a continuation never shares state with the launch beside it.  Robots do not interact, so
    a robot's sequence of solves - and with it every number it produces - is the same as in lock step.
"""
import numpy as np


class DeviceFleet:
    def __init__(self, mm, x0, glob, obs0, vel, N=30, device=0, dt=0.1, handles=3):
        import torch
        self.torch = torch
        self.B, self.N, self.M, self.dt = int(x0.shape[0]), int(N), int(obs0.shape[1]), float(dt)
        self.dev = torch.device("cuda", device)
        B, M = self.B, self.M
        mk = lambda: mm.MPCWholeBody(mm.MobileManipulator(dt), [], [], N=N, max_batch=max(B, 1), device=device, n_obstacles=M, obs_per_stage=True)
        # [0]: lock step (plain kernel); [1], [2]: the two alternating handles of run_async, created on its first call.  A handle owns
        # max_batch rows of device state: staging, warm start, the second-order-correction scratch and - long horizons - the gain
        # blocks (mmpc_create: about 0.5 MB per robot at N = 30, M = 8)
        self._mk = mk
        self.ctrls = [mk()]
        self.engs = [c._engine for c in self.ctrls]
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.f64 = f64
        as_t = lambda a: a.to(self.dev) if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.x0, self.glob, self.obs0, self.vel = as_t(x0), as_t(glob), as_t(obs0), as_t(vel)
        self.nglob = self.glob.shape[1]
        self.xlo = torch.from_numpy(self.ctrls[0].xlim[0]).to(self.dev); self.xhi = torch.from_numpy(self.ctrls[0].xlim[1]).to(self.dev)
        self.uref = torch.zeros((B, N, 5), **f64)
        self.karr = torch.arange(N + 1, **f64)
        self.rows = torch.arange(B, device=self.dev)

    # ---- per-robot pieces (torch ops on the device)
    def plant(self, x, u):
        torch = self.torch
        xc = torch.minimum(torch.maximum(x, self.xlo), self.xhi)             # solve() clips x_init (:290-291); the driver steps the clipped state
        c, s, dt = torch.cos(xc[:, 2]), torch.sin(xc[:, 2]), self.dt
        return torch.stack([xc[:, 0] + dt * xc[:, 3], xc[:, 1] + dt * xc[:, 4], xc[:, 2] + dt * xc[:, 5],
                            xc[:, 3] + dt * (u[:, 0] * c - xc[:, 4] * xc[:, 5]),
                            xc[:, 4] + dt * (u[:, 0] * s + xc[:, 3] * xc[:, 5]), xc[:, 5] + dt * u[:, 1],
                            xc[:, 6] + dt * u[:, 2], xc[:, 7] + dt * u[:, 3], xc[:, 8] + dt * u[:, 4]], dim=1)

    def inputs(self, x, tick, loc_out=None, obs_out=None):
        """local reference window (nearest point of the global plan, tail padded: interface_wholebody_qref.py:377-389) and the
        per-stage obstacle table of every robot at ITS tick (a (B,) tensor)."""
        torch = self.torch
        B, N = self.B, self.N
        d = torch.linalg.norm(x[:, None, :2] - self.glob[:, :, :2], dim=2)
        start = torch.argmin(d, dim=1)
        idx = torch.clamp(start[:, None] + torch.arange(N + 1, device=self.dev)[None, :], max=self.nglob - 1)
        loc = torch.gather(self.glob, 1, idx[:, :, None].expand(B, N + 1, 9))
        obs = self.obs0[:, None, :, :].repeat(1, N + 1, 1, 1)
        tk = (tick.to(torch.float64)[:, None] + self.karr[None, :]) * self.dt                   # (B, N+1)
        obs[..., :2] += self.vel[:, None, :, :] * tk[:, :, None, None]
        if loc_out is not None:
            loc_out.copy_(loc); obs_out.copy_(obs)
            return loc_out, obs_out
        return loc.contiguous(), obs.contiguous()

    # ---- lock step
    def run_lockstep(self, T):
        res = {}
        for _ in self._lockstep_ticks(T, res):
            pass
        return res

    def _lockstep_ticks(self, T, res, on_tick=None):
        """run_lockstep as a generator: one tick's work is queued on the current stream per next(); the result lands in `res`.
        on_tick(t, u0): called after tick t's solve is queued, on the stream it runs on, with the (B, 5) first inputs - where a
        multi-rank run issues its all-gather of u0 (bench.py --config c5)"""
        torch = self.torch
        B, N = self.B, self.N
        eng = self.engs[0]
        eng.set_iteration_budget(0); eng.reset()
        x = self.x0.clone(); ul = torch.zeros((B, N, 5), **self.f64)
        hist = torch.zeros((B, T, 5), **self.f64); its = torch.zeros((B, T), dtype=torch.int32, device=self.dev)
        tick = torch.zeros(B, dtype=torch.int64, device=self.dev)
        out = None
        ok = torch.ones((), dtype=torch.bool, device=self.dev)
        for t in range(T):
            loc, obs = self.inputs(x, tick)
            out = eng.solve_batch_device(x, loc, self.uref, ul, obs, out=out)
            ok = ok & (out["status"] == 0).all()
            ul = out["U"].clone()
            u0 = out["U"][:, 0]
            hist[:, t] = u0; its[:, t] = out["iters"]
            if on_tick is not None:
                on_tick(t, u0)
            x = self.plant(x, u0)
            tick += 1
            res.update(u0=hist, x=x, iters=its, all_converged=ok, rounds=T)
            yield t

    # ---- groups in lock step, out of phase
    def run_groups(self, T, groups=2, priority=True, on_tick=None):
        """The fleet as `groups` contiguous groups of robots, each in lock step on its own HIP stream and handle, the streams at
        descending priority: the workgroup dispatcher serves the first group's launch first and fills the slots its tail leaves
        empty - a tick lasts as long as its slowest robot, 200-450 iterations against a mean of 34 - with the other group's
        launch, so the groups run out of phase and one group's tail overlaps the other's bulk.  Robots do not interact: every
        robot's numbers are those of run_lockstep (tests/test_gpu_parity.py).  Same reference protocol per robot.
        on_tick(g, lo, hi, t, u0): called per group and tick on the group's stream (see _lockstep_ticks) - a multi-rank run gathers
        the first inputs of group g's robots [lo, hi) there, so that a tick's exchange waits for the slowest robot of ONE group of
        one rank, not of the whole node."""
        torch = self.torch
        G = int(groups)
        if not hasattr(self, "_groups") or len(self._groups) != G or getattr(self, "_groups_pr", True) != bool(priority):
            self._groups_pr = bool(priority)
            from . import sharding
            import sys
            mm = sys.modules[__package__]
            lo_pr, hi_pr = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            self._groups = []
            for g in range(G):
                lo, hi = sharding.shard_bounds(self.B, G, g)
                sub = DeviceFleet(mm, self.x0[lo:hi], self.glob[lo:hi], self.obs0[lo:hi], self.vel[lo:hi], N=self.N, device=self.dev.index, dt=self.dt,
                                  handles=1)
                # (priority: lower number = served first; the last group runs at the default priority)
                pr = (max(hi_pr, min(lo_pr, 0 - (G - 1 - g))) if hi_pr < 0 else 0) if priority else 0
                self._groups.append((lo, hi, sub, torch.cuda.Stream(device=self.dev, priority=pr)))
        main = torch.cuda.current_stream(self.dev)
        ev0 = torch.cuda.Event(); ev0.record(main)
        res = [dict() for _ in range(G)]
        gens = []
        for g, (lo, hi, sub, st) in enumerate(self._groups):
            st.wait_event(ev0)
            with torch.cuda.stream(st):
                cb = None if on_tick is None else (lambda t, u0, g=g, lo=lo, hi=hi: on_tick(g, lo, hi, t, u0))
                gens.append(sub._lockstep_ticks(T, res[g], cb))
        for t in range(T):
            for g, (lo, hi, sub, st) in enumerate(self._groups):
                with torch.cuda.stream(st):
                    next(gens[g])
        for g, (lo, hi, sub, st) in enumerate(self._groups):
            ev = torch.cuda.Event(); ev.record(st); main.wait_event(ev)
            for v in res[g].values():        # (allocated on the group's stream, read on the caller's from here on)
                if torch.is_tensor(v):
                    v.record_stream(main)
        ok = res[0]["all_converged"]
        for r_ in res[1:]:
            ok = ok & r_["all_converged"]
        return dict(u0=torch.cat([r_["u0"] for r_ in res]), x=torch.cat([r_["x"] for r_ in res]), iters=torch.cat([r_["iters"] for r_ in res]),
                    all_converged=ok, rounds=T, groups=G)

    # ---- asynchronous
    def run_async(self, T, budget=48, max_rounds=None):
        torch = self.torch
        B, N, M = self.B, self.N, self.M
        dev = self.dev
        main = torch.cuda.current_stream(dev)
        while len(self.ctrls) < 3:
            self.ctrls.append(self._mk()); self.engs.append(self.ctrls[-1]._engine)
        if not hasattr(self, "_sets"):
            self._sides = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            self._sets = []
            for s in range(2):
                o = dict(X=torch.zeros((B, N + 1, 9), **self.f64), U=torch.zeros((B, N, 5), **self.f64), s=torch.zeros((B, N + 1), **self.f64),
                         status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev),
                         cost=torch.zeros(B, **self.f64), err=torch.zeros(B, **self.f64))
                self._sets.append(dict(out=o, x_in=torch.zeros((B, 9), **self.f64), loc=torch.zeros((B, N + 1, 9), **self.f64),
                                       obs=torch.zeros((B, N + 1, M, 3), **self.f64), ul_in=torch.zeros((B, N, 5), **self.f64),
                                       list=torch.zeros(B, dtype=torch.int32, device=dev), count=torch.zeros(1, dtype=torch.int32, device=dev),
                                       susp=torch.zeros(B, dtype=torch.bool, device=dev), ev=None))
        for s in range(2):
            self.engs[1 + s].set_iteration_budget(budget); self.engs[1 + s].reset()
            self._sets[s]["ev"] = None; self._sets[s]["susp"].zero_()
        x = self.x0.clone(); ul = torch.zeros((B, N, 5), **self.f64)
        hist = torch.zeros((B, T, 5), **self.f64); its = torch.zeros((B, T), dtype=torch.int32, device=dev)
        tick = torch.zeros(B, dtype=torch.int64, device=dev)
        inflight = torch.zeros(B, dtype=torch.bool, device=dev)
        prev_it = torch.zeros(B, **self.f64)
        failed = torch.zeros((), dtype=torch.bool, device=dev)
        nsusp = torch.zeros((), dtype=torch.int64, device=dev)
        state = dict(x=x, ul=ul, tick=tick, prev_it=prev_it)

        def advance(mask, out):
            # the robots of `mask` take the solution in `out`: u_latest <- U*, plant step with u0, next tick
            u0 = out["U"][:, 0]
            t_idx = state["tick"].clamp(max=T - 1)
            hist[self.rows, t_idx] = torch.where(mask[:, None], u0, hist[self.rows, t_idx])
            its[self.rows, t_idx] = torch.where(mask, out["iters"], its[self.rows, t_idx])
            state["x"] = torch.where(mask[:, None], self.plant(state["x"], u0), state["x"])
            state["ul"] = torch.where(mask[:, None, None], out["U"], state["ul"])
            state["prev_it"] = torch.where(mask, out["iters"].to(torch.float64), state["prev_it"])
            state["tick"] = state["tick"] + mask.to(torch.int64)

        r = 0
        cap = max_rounds if max_rounds is not None else 4 * T + 8
        while True:
            s = r & 1
            S = self._sets[s]
            eng = self.engs[1 + s]
            if S["ev"] is not None:
                # the continuation of round r-2 (same handle, same buffers) is finished: its robots rejoin, one tick behind
                main.wait_event(S["ev"]); S["ev"] = None
                m = S["susp"]
                failed = failed | (m & (S["out"]["status"] != 0)).any()
                advance(m, S["out"])
                inflight = inflight & ~m
            ready = (~inflight) & (state["tick"] < T)
            S["x_in"].copy_(state["x"]); S["ul_in"].copy_(state["ul"])
            self.inputs(state["x"], state["tick"], S["loc"], S["obs"])
            score = torch.where(ready, state["prev_it"] + 1.0, torch.zeros_like(state["prev_it"]))
            S["list"].copy_(torch.argsort(score, descending=True, stable=True).to(torch.int32))
            S["count"].copy_(ready.sum().to(torch.int32).reshape(1))
            eng.solve_batch_device(S["x_in"], S["loc"], self.uref, S["ul_in"], S["obs"], out=S["out"], rows=(S["list"], S["count"]))
            # (snapshots before the continuation on the side stream starts rewriting the rows of the suspended robots: advance() below
            #  reads whole columns on the main stream)
            snap = S["out"]["status"].clone()
            snap_out = dict(U=S["out"]["U"].clone(), iters=S["out"]["iters"].clone())
            evm = torch.cuda.Event(); evm.record(main)
            side = self._sides[s]
            side.wait_event(evm)
            eng.resume_batch_device(S["x_in"], S["loc"], self.uref, S["ul_in"], S["obs"], S["out"], stream=side.cuda_stream)
            S["ev"] = torch.cuda.Event(); S["ev"].record(side)
            conv = ready & (snap == 0)
            susp = ready & (snap == 3)
            failed = failed | (ready & (snap != 0) & (snap != 3)).any()
            nsusp = nsusp + susp.sum()
            advance(conv, snap_out)
            inflight = inflight | susp
            S["susp"] = susp
            r += 1
            if r >= T and (r >= cap or not bool(((state["tick"] < T) | inflight).any())):
                break
        for s in range(2):
            if self._sets[s]["ev"] is not None:
                self._sets[s]["ev"].synchronize(); self._sets[s]["ev"] = None
            self.engs[1 + s].set_iteration_budget(0)
        done = bool((state["tick"] >= T).all())
        return dict(u0=hist, x=state["x"], iters=its, all_converged=(~failed) & torch.tensor(done, device=dev), rounds=r, suspended=nsusp)
