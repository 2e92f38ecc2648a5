"""ctypes binding of the C ABI in include/mmpc.h (csrc/libmmpc.so).  No fallback of any kind:
a missing library or a failing call raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMPC_LIB") or os.path.join(_HERE, "csrc", "libmmpc.so")   # MMPC_LIB: A/B builds of the same HIP library

KIND_WHOLEBODY, KIND_BASE, KIND_WHOLEBODY_POSE = 0, 1, 2
STATUS_CONVERGED, STATUS_MAXITER, STATUS_NUMERIC, STATUS_SUSPENDED = 0, 1, 2, 3
STATUS_Q8_REFUSED = 8      # host-side only (controllers/_q8.py): converged, but an as-written extra half-space row is violated


class MmpcConfig(C.Structure):
    _fields_ = [("kind", C.c_int), ("N", C.c_int), ("M", C.c_int), ("obs_per_stage", C.c_int),
                ("max_batch", C.c_int), ("device", C.c_int), ("max_iter", C.c_int),
                ("dt", C.c_double), ("tol", C.c_double), ("mu_init", C.c_double),
                ("ulim", (C.c_double * 5) * 2), ("xlim", (C.c_double * 9) * 2), ("dulim", (C.c_double * 5) * 2),
                ("L", C.c_int), ("halfspace", (C.c_double * 6) * 8), ("as_written", C.c_int)]


EXPORTS = ["mmpc_create", "mmpc_destroy", "mmpc_set_weights", "mmpc_set_terminal_xy_equality", "mmpc_reset",
           "mmpc_solve_batch", "mmpc_solve_batch_device", "mmpc_get_u_latest", "mmpc_set_u_latest",
           "mmpc_lds_bytes", "mmpc_problems_per_cu", "mmpc_set_warm_start", "mmpc_set_schedule_hint", "mmpc_set_iteration_budget", "mmpc_resume_batch_device", "mmpc_solve_list_device", "mmpc_suspended_count", "mmpc_last_error", "mmpc_version", "mmpc_ik_batch", "mmpc_ik_batch_device"]

_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib():
    """Loads libmmpc.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "HIP extension %s is missing - build it with __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        # torch ships its own libamdhip64.so.7 / libhsa-runtime64.so.1; libmmpc.so asks for the same sonames.  Whichever is
        # mapped first serves both, and torch cannot initialise on the system copy ("No HIP GPUs are available" when the first
        # torch.cuda call comes after a solve) - so the process settles on torch's runtime before the library is opened.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.mmpc_create.argtypes = [C.POINTER(MmpcConfig), C.POINTER(C.c_void_p)]
        L.mmpc_destroy.argtypes = [C.c_void_p]
        L.mmpc_set_weights.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double, _dp]
        L.mmpc_set_terminal_xy_equality.argtypes = [C.c_void_p, C.c_int]
        L.mmpc_reset.argtypes = [C.c_void_p]
        L.mmpc_solve_batch.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp]
        L.mmpc_solve_batch_device.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 13 + [C.c_void_p]
        L.mmpc_get_u_latest.argtypes = [C.c_void_p, C.c_int, _dp]
        L.mmpc_set_u_latest.argtypes = [C.c_void_p, C.c_int, _dp]
        L.mmpc_lds_bytes.argtypes = [C.c_void_p]
        L.mmpc_problems_per_cu.argtypes = [C.c_void_p]
        L.mmpc_set_warm_start.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        L.mmpc_set_schedule_hint.argtypes = [C.c_void_p, C.c_int]
        L.mmpc_set_iteration_budget.argtypes = [C.c_void_p, C.c_int]
        L.mmpc_resume_batch_device.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 13 + [C.c_void_p]
        L.mmpc_solve_list_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 13 + [C.c_void_p]
        L.mmpc_suspended_count.argtypes = [C.c_void_p, _ip]
        L.mmpc_last_error.argtypes = [C.c_void_p]
        L.mmpc_last_error.restype = C.c_char_p
        L.mmpc_version.restype = C.c_char_p
        L.mmpc_ik_batch.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _ip, _ip]
        L.mmpc_ik_batch_device.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_void_p]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def ik_batch(q0, target_xz, device=0):
    """mmpc_ik_batch: q0 (B,3), target_xz (B,2) -> dict(q (B,3), status (B,), iters (B,)).  Runs on the GPU (no fallback)."""
    q0 = np.ascontiguousarray(np.atleast_2d(q0), np.float64)
    t = np.ascontiguousarray(np.atleast_2d(target_xz), np.float64)
    B = q0.shape[0]
    if q0.shape != (B, 3) or t.shape != (B, 2):
        raise ValueError("q0 must be (B,3) and target_xz (B,2)")
    q = np.empty((B, 3)); st = np.empty(B, np.int32); it = np.empty(B, np.int32)
    L = lib()
    rc = L.mmpc_ik_batch(int(device), B, _d(q0), _d(t), _d(q), _i(st), _i(it))
    if rc != 0:
        raise RuntimeError("mmpc_ik_batch failed (%d): %s" % (rc, (L.mmpc_last_error(None) or b"").decode()))
    return dict(q=q, status=st, iters=it)


class Engine:
    """Owns one mmpc_handle (one controller's NLP structure on one GPU)."""

    def __init__(self, kind, N, M, dt, ulim, xlim, dulim, max_batch=1, device=0, obs_per_stage=False,
                 tol=1e-8, mu_init=1.0, max_iter=2000, halfspaces=None, as_written=False):
        self.kind, self.N, self.M = kind, int(N), int(M)
        self.nx, self.nu = (6, 2) if kind == KIND_BASE else (9, 5)
        self.nref = 4 if kind == KIND_WHOLEBODY_POSE else self.nx      # reference row: endpoint pose (x,y,z,psi) or the state
        self.max_batch, self.device, self.obs_per_stage = int(max_batch), int(device), bool(obs_per_stage)
        cfg = MmpcConfig()
        cfg.kind, cfg.N, cfg.M, cfg.obs_per_stage = kind, self.N, self.M, int(self.obs_per_stage)
        cfg.max_batch, cfg.device, cfg.max_iter = self.max_batch, self.device, int(max_iter)
        cfg.dt, cfg.tol, cfg.mu_init = float(dt), float(tol), float(mu_init)
        ulim = np.asarray(ulim, float); xlim = np.asarray(xlim, float); dulim = np.asarray(dulim, float)
        for r in range(2):
            for j in range(5):
                cfg.ulim[r][j] = ulim[r, j] if j < self.nu else 0.0
                cfg.dulim[r][j] = dulim[r, j] if j < self.nu else 0.0
            for j in range(9):
                cfg.xlim[r][j] = xlim[r, j] if j < self.nx else 0.0
        cfg.L = 0
        cfg.as_written = int(bool(as_written))
        if halfspaces is not None and len(halfspaces):
            hs = np.asarray(halfspaces, float).reshape(-1, 6)
            if hs.shape[0] > 8:
                raise ValueError("at most 8 half-space obstacles")
            cfg.L = hs.shape[0]
            for j in range(cfg.L):
                for a in range(6):
                    cfg.halfspace[j][a] = hs[j, a]
        self._h = C.c_void_p()
        rc = lib().mmpc_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = lib().mmpc_last_error(self._h).decode() if self._h else ""
            if self._h:
                lib().mmpc_destroy(self._h)
                self._h = C.c_void_p()
            raise RuntimeError("mmpc_create failed (%d) %s" % (rc, msg))

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, lib().mmpc_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None):
            lib().mmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def lds_bytes(self):
        return lib().mmpc_lds_bytes(self._h)

    @property
    def problems_per_cu(self):
        return lib().mmpc_problems_per_cu(self._h)

    def set_weights(self, Q=None, R=None, P=None, S=None, W=None):
        def m(a, n):
            if a is None:
                return None
            a = np.ascontiguousarray(a, float)
            if a.shape != (n, n):
                raise ValueError("weight matrix must be %dx%d" % (n, n))
            return a
        Q, P, R, W = m(Q, self.nref), m(P, self.nref), m(R, self.nu), m(W, self.nu)
        Sv = -1.0 if S is None else float(np.ravel(S)[0])
        self._chk(lib().mmpc_set_weights(self._h, _d(Q), _d(R), _d(P), Sv, _d(W)), "mmpc_set_weights")

    def set_warm_start(self, u_guess=None, mu_init=1.0):
        """mmpc_set_warm_start: `u_guess` (B,N,nu) cuda float64 tensor (kept alive by the engine) or None, initial barrier
        parameter `mu_init`.  Applies to the following solve_batch_device calls (their x_guess is the initial X)."""
        if u_guess is not None:
            import torch
            if (not u_guess.is_cuda or u_guess.dtype != torch.float64 or not u_guess.is_contiguous()
                    or u_guess.device.index != self.device or tuple(u_guess.shape[1:]) != (self.N, self.nu)):
                raise ValueError("u_guess must be a contiguous cuda:%d float64 tensor (B, N, nu)" % self.device)
        self._u_guess = u_guess
        self._chk(lib().mmpc_set_warm_start(self._h, C.c_void_p(u_guess.data_ptr()) if u_guess is not None else None,
                                            float(mu_init)), "mmpc_set_warm_start")

    def set_schedule_hint(self, mode):
        """mmpc_set_schedule_hint: launch order of a batch's workgroups - 0/False batch order, 1/True (default) longest-first
        by the previous launch's iteration counts when there is one, else by the a-priori difficulty key of the batch's own
        data, 2 always the a-priori key."""
        self._chk(lib().mmpc_set_schedule_hint(self._h, int(mode)), "mmpc_set_schedule_hint")

    def set_iteration_budget(self, budget):
        """mmpc_set_iteration_budget: iterations a launch may spend on an instance before it is suspended (0: off)."""
        self._chk(lib().mmpc_set_iteration_budget(self._h, int(budget)), "mmpc_set_iteration_budget")

    def suspended_count(self):
        """mmpc_suspended_count: instances the last budgeted launch left suspended (waits for the handle's launches)."""
        n = C.c_int(0)
        self._chk(lib().mmpc_suspended_count(self._h, C.byref(n)), "mmpc_suspended_count")
        return n.value

    def set_terminal_xy_equality(self, on):
        self._chk(lib().mmpc_set_terminal_xy_equality(self._h, int(bool(on))), "mmpc_set_terminal_xy_equality")

    def reset(self):
        self._chk(lib().mmpc_reset(self._h), "mmpc_reset")

    def obs_shape(self, B):
        return (B, self.N + 1, self.M, 3) if self.obs_per_stage else (B, self.M, 3)

    def solve_batch(self, x_init, traj_ref, u_ref, obs):
        """Host arrays in, host arrays out (dict).  Updates the handle's warm start."""
        N, nx, nu = self.N, self.nx, self.nu
        x_init = np.ascontiguousarray(x_init, float)
        B = x_init.shape[0]
        traj_ref = np.ascontiguousarray(traj_ref, float)
        u_ref = np.ascontiguousarray(u_ref, float)
        obs = np.ascontiguousarray(obs, float)
        if x_init.shape != (B, nx) or traj_ref.shape != (B, N + 1, self.nref) or u_ref.shape != (B, N, nu):
            raise ValueError("shape mismatch: x_init %s traj_ref %s u_ref %s" % (x_init.shape, traj_ref.shape, u_ref.shape))
        if obs.shape != self.obs_shape(B):
            raise ValueError("obs shape %s, expected %s" % (obs.shape, self.obs_shape(B)))
        out = dict(u0=np.empty((B, nu)), X=np.empty((B, N + 1, nx)), U=np.empty((B, N, nu)), s=np.empty((B, N + 1)),
                   status=np.empty(B, np.int32), iters=np.empty(B, np.int32), cost=np.empty(B))
        self._chk(lib().mmpc_solve_batch(self._h, B, _d(x_init), _d(traj_ref), _d(u_ref), _d(obs), _d(out["u0"]),
                                         _d(out["X"]), _d(out["U"]), _d(out["s"]), _i(out["status"]),
                                         _i(out["iters"]), _d(out["cost"])), "mmpc_solve_batch")
        return out

    def get_u_latest(self, B):
        a = np.empty((B, self.N, self.nu))
        self._chk(lib().mmpc_get_u_latest(self._h, B, _d(a)), "mmpc_get_u_latest")
        return a

    def set_u_latest(self, u):
        u = np.ascontiguousarray(u, float)
        self._chk(lib().mmpc_set_u_latest(self._h, u.shape[0], _d(u)), "mmpc_set_u_latest")

    def solve_batch_device(self, x_init, traj_ref, u_ref, u_last, obs, x_guess=None, out=None, stream=None, resume=False, rows=None):
        """torch CUDA float64 tensors in / out, asynchronous on torch's current stream (or `stream`).
        PyTorch is only the owner of the device memory and of the stream here.
        rows = (list, count): mmpc_solve_list_device - only the instances list[0 .. count[0]-1] (int32 cuda tensors: a list of
        row indices and a one-element count, both read on the device) are solved, in list order; the other rows are not touched."""
        import torch
        if rows is not None and (out is None or resume):
            raise ValueError("a list launch needs the output set of the whole batch (out=...) and is not a continuation")
        N, nx, nu = self.N, self.nx, self.nu
        B = x_init.shape[0]
        for t_, shp in ((x_init, (B, nx)), (traj_ref, (B, N + 1, self.nref)), (u_ref, (B, N, nu)), (u_last, (B, N, nu))):
            if tuple(t_.shape) != shp or t_.dtype != torch.float64 or not t_.is_cuda or not t_.is_contiguous():
                raise ValueError("device tensor must be contiguous cuda float64 of shape %s" % (shp,))
        if tuple(obs.shape) != self.obs_shape(B) or obs.dtype != torch.float64 or not obs.is_cuda or not obs.is_contiguous():
            raise ValueError("obs must be contiguous cuda float64 of shape %s" % (self.obs_shape(B),))
        if x_guess is not None and (tuple(x_guess.shape) != (B, N + 1, nx) or x_guess.dtype != torch.float64
                                    or not x_guess.is_cuda or not x_guess.is_contiguous()):
            raise ValueError("x_guess must be contiguous cuda float64 of shape %s" % ((B, N + 1, nx),))
        dev = x_init.device
        for t_ in (traj_ref, u_ref, u_last, obs, x_guess):
            if t_ is not None and t_.device != dev:
                raise ValueError("all device tensors of one call must live on %s" % dev)
        if getattr(self, "_u_guess", None) is not None and self._u_guess.shape[0] < B:
            raise ValueError("the u_guess set by set_warm_start holds %d instances, this call has %d" % (self._u_guess.shape[0], B))
        if out is None:
            out = dict(X=torch.empty((B, N + 1, nx), dtype=torch.float64, device=dev),
                       U=torch.empty((B, N, nu), dtype=torch.float64, device=dev),
                       s=torch.empty((B, N + 1), dtype=torch.float64, device=dev),
                       status=torch.empty(B, dtype=torch.int32, device=dev),
                       iters=torch.empty(B, dtype=torch.int32, device=dev),
                       cost=torch.empty(B, dtype=torch.float64, device=dev),
                       err=torch.empty(B, dtype=torch.float64, device=dev))
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        p = lambda t_: C.c_void_p(t_.data_ptr()) if t_ is not None else None
        if rows is not None:
            lst, cnt = rows
            for t_ in (lst, cnt):
                if t_.dtype != torch.int32 or not t_.is_cuda or not t_.is_contiguous() or t_.device != dev:
                    raise ValueError("rows = (list, count): contiguous int32 tensors on %s" % dev)
            if cnt.numel() != 1 or lst.numel() < 1:
                raise ValueError("rows = (list, count): a one-element count and a list of at least one entry")
            self._chk(lib().mmpc_solve_list_device(self._h, B, p(lst), p(cnt), int(min(lst.numel(), B)), p(x_init), p(traj_ref), p(u_ref), p(u_last),
                                                   p(x_guess), p(obs), p(out["X"]), p(out["U"]), p(out["s"]), p(out["status"]),
                                                   p(out["iters"]), p(out["cost"]), p(out["err"]), C.c_void_p(st)), "mmpc_solve_list_device")
            return out
        fn = lib().mmpc_resume_batch_device if resume else lib().mmpc_solve_batch_device
        self._chk(fn(self._h, B, p(x_init), p(traj_ref), p(u_ref), p(u_last), p(x_guess),
                     p(obs), p(out["X"]), p(out["U"]), p(out["s"]), p(out["status"]),
                     p(out["iters"]), p(out["cost"]), p(out["err"]), C.c_void_p(st)),
                  "mmpc_resume_batch_device" if resume else "mmpc_solve_batch_device")
        return out

    def resume_batch_device(self, x_init, traj_ref, u_ref, u_last, obs, out, x_guess=None, stream=None):
        """mmpc_resume_batch_device: continues the instances the preceding budgeted solve_batch_device call (same tensors,
        same `out`) left suspended; the other rows of `out` are not touched."""
        return self.solve_batch_device(x_init, traj_ref, u_ref, u_last, obs, x_guess=x_guess, out=out, stream=stream, resume=True)
