"""C4 batch under the three launch orders (batch order / a-priori key / exact history) - ms per launch for the library in MMPC_LIB."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, mmpc_loader
mm = mmpc_loader.load()
synth = mm.synth
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B, N, M = 8192, 20, 5
d = synth.make_batch(B, N=N, M=M, config_id=seed)
dev = torch.device("cuda", 0)
ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=B, n_obstacles=M)
eng = ctrl._engine
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
xi = t(np.clip(d["x_init"], ctrl.xlim[0], ctrl.xlim[1])); tr = t(d["traj_ref"]); ur = t(d["u_ref"]); ob = t(d["obs"])
ul = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
res = []
for mode, name in ((0, "batch order"), (2, "a-priori key"), (1, "history (exact)")):
    eng.set_schedule_hint(mode)
    for _ in range(3): out = eng.solve_batch_device(xi, tr, ur, ul, ob)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = eng.solve_batch_device(xi, tr, ur, ul, ob)
    torch.cuda.synchronize(); res.append("%s %.3f ms" % (name, (time.perf_counter() - t0) * 100))
it = out["iters"].cpu().numpy()
print(os.path.basename(os.environ.get("MMPC_LIB", "libmmpc.so")), "seed", seed, "|", " | ".join(res), "| iters mean %.2f max %d sum %d" % (it.mean(), it.max(), it.sum()))
