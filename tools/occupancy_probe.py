"""Two resident waves per SIMD, measured on the shipped kernel code at half the horizon: the (0, 10, 5) instantiation of
mmpc_fast_kernel needs 21.4 KB of LDS per problem (7 problems per CU), so its occupancy is set by the register allocation:
libmmpc_n10w1.so (__launch_bounds__(64, 1): 466 registers, 4 problems per CU, one wave per SIMD) against libmmpc_n10w2.so
(__launch_bounds__(64, 2): 256 registers + scratch spills, 7 per CU).  Built by tools/occupancy_probe.sh; MMPC_LIB selects."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, mmpc_loader
from oracle import synth
mm = mmpc_loader.load()
N, M, B = 10, 5, 8192
d = synth.make_batch(B, N=N, M=M, config_id=11)
dev = torch.device("cuda", 0)
ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=B, n_obstacles=M)
eng = ctrl._engine
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
xi = t(np.clip(d["x_init"], ctrl.xlim[0], ctrl.xlim[1])); tr = t(d["traj_ref"]); ur = t(d["u_ref"]); ob = t(d["obs"])
ul = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
out = eng.solve_batch_device(xi, tr, ur, ul, ob); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(8):
    e0.record(); eng.solve_batch_device(xi, tr, ur, ul, ob, out=out); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
it = out["iters"].cpu().numpy()
print("%s: N=10 M=5 B=%d lds %d B, %d per CU (%s): %.3f ms (median %.3f), iters mean %.2f max %d, conv %.4f, checksum %.9f" % (
    os.path.basename(os.environ.get("MMPC_LIB", "libmmpc.so")), B, eng.lds_bytes, eng.problems_per_cu, eng.kernel_name if hasattr(eng, "kernel_name") else "",
    min(ts), float(np.median(ts)), it.mean(), it.max(), (out["status"] == 0).float().mean().item(), float(out["X"].double().sum().item())))
