"""Per-phase wave-cycle shares of the fast kernel (diagnostic build libmmpc_stamp.so, -DMMPC_STAMP)."""
import sys, os, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["MMPC_LIB"] = os.environ.get("MMPC_STAMP_LIB", os.path.join(ROOT, "mobile-manipulator-mpc_amd", "csrc", "libmmpc_stamp.so"))
import torch, mmpc_loader
from oracle import synth
mm = mmpc_loader.load()
N, M, B = int(os.environ.get("MMPC_PROBE_N", 20)), int(os.environ.get("MMPC_PROBE_M", 5)), 1024   # (30, 8: the C5 kernel, obstacle table per stage)
OPS = N == 30
d = synth.make_batch(8192, N=N, M=M, config_id=5, moving=True) if OPS else synth.make_batch(8192, N=N, M=M)
if OPS: d["obs"] = d["obs"][:, None, :, :] + np.arange(N + 1)[None, :, None, None] * 0.1 * np.concatenate([d["obs_vel"], np.zeros_like(d["obs_vel"][..., :1])], axis=-1)[:, None, :, :]
dev = torch.device("cuda", 0)
ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=B, n_obstacles=M, obs_per_stage=OPS)
eng = ctrl._engine
idx = np.arange(B) if os.environ.get("MMPC_PROBE_DISTINCT") else np.zeros(B, int)   # (distinct instances: the waves of a CU drift out of phase)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a[idx])).to(dev)
xi = t(np.clip(d["x_init"], ctrl.xlim[0], ctrl.xlim[1])); tr = t(d["traj_ref"]); ur = t(d["u_ref"]); ob = t(d["obs"])
ul = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
out = eng.solve_batch_device(xi, tr, ur, ul, ob); torch.cuda.synchronize()
L = mm._capi.lib()
buf = (ctypes.c_ulonglong * 16)()
L.mmpc_debug_read_stamps(buf)
out = eng.solve_batch_device(xi, tr, ur, ul, ob); torch.cuda.synchronize()
L.mmpc_debug_read_stamps(buf)
v = np.array(list(buf), float)
names = ["line search: move to the trial point", "evaluation, stage lanes", "convergence / barrier update", "A1 stage", "A1 pair",
         "R0 / P store + operand hand-over", "R1+R2 (T, M MFMA chains)", "R3 (input elimination: 5 rank-one MFMAs)", "gain back-substitution", "forward", "D1", "D2 (row steps)",
         "evaluation, pair lanes + filter test", "exit"]
tot = v.sum()
it = out["iters"].float().mean().item()
for i, n in enumerate(names):
    # stamp i closes the segment that ENDS at MMPC_TS(i): segment i = code between TS(i-1) and TS(i)
    print("%-22s %6.1f %%   %8.0f cycles/iter/wave" % ("-> " + n, 100 * v[i] / tot, v[i] / B / it))
print("total cycles/iter/wave %.0f  (iters %.1f)" % (tot / B / it, it))
