"""Generic kernel on the demo's shape (C1: N = 20, three circle obstacles, two half-space planes, rows as written): batch of
starts around the 'tent' (as tests/test_gpu_certificates.py::test_as_written_halfspace_rows_batch), device-resident timing."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, mmpc_loader
mm = mmpc_loader.load()
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 20
as_written = "--intended" not in sys.argv
r2 = 1 / np.sqrt(2)
hs = np.array([[2.5, 2, 0.35 + 0.606 + 0.333, r2, 0, r2], [2.5, 2, 0.35 + 0.606 + 0.333, -r2, 0, r2]])
rng = np.random.default_rng(11)
x = np.zeros((B, 9)); tr = np.zeros((B, N + 1, 9))
for b in range(B):
    x0 = np.array([rng.uniform(1.4, 2.6), rng.uniform(1.6, 2.4), rng.uniform(-0.4, 0.4), rng.uniform(0, 0.8), 0, 0,
                   rng.uniform(-0.3, 0.6), rng.uniform(-1.6, -0.6), rng.uniform(0.8, 2.2)])
    x0[4] = x0[3] * np.sin(x0[2]); x0[3] = x0[3] * np.cos(x0[2])
    tg = x0.copy(); tg[0] += rng.uniform(0.8, 1.8); tg[1] += rng.uniform(-0.3, 0.3); tg[3:6] = 0
    x[b] = x0; tr[b] = np.linspace(x0, tg, 51)[:N + 1]
obs = np.broadcast_to(np.array([[2.5, 3.4, 0.3], [2.5, 0.6, 0.3], [6, 6, 0.1]]), (B, 3, 3)).copy()
oml = [(h[:3], h[3:].reshape(1, 3)) for h in hs]
ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], oml, N=N, max_batch=B, n_obstacles=3, faithful_convex=None if as_written else False)
eng = ctrl._engine
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
xi, trd, ob = t(x), t(tr), t(obs)
z = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
eng.set_schedule_hint(1 if "--history" in sys.argv else 2)
out = eng.solve_batch_device(xi, trd, z, z, ob); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(4):
    e0.record(); eng.solve_batch_device(xi, trd, z, z, ob, out=out); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
it = out["iters"].cpu().numpy(); st = out["status"].cpu().numpy()
print("C1 shape (%s rows) B=%d lds %d B, %d per CU: %.2f ms -> %.0f solves/s; iters mean %.1f max %d; converged %.4f; checksum %.9f" % (
    "as-written" if as_written else "intended", B, eng.lds_bytes, eng.problems_per_cu, min(ts), B / min(ts) * 1e3, it.mean(), it.max(), (st == 0).mean(),
    float(out["X"][st == 0].double().sum().item())))

if "--dump-iters" in sys.argv:
    np.save(os.path.join(ROOT, "gpurun_out", "c1_iters_%s_%d.npy" % ("aw" if as_written else "int", B)), it)
