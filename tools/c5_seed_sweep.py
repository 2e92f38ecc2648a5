"""C5 fleets of other seeds (the ranks of an 8-GPU weak-scaling run simulate seeds 5 ... 12): ten ticks in lock step and in three groups
out of phase - every solve converged?  slowest robot per tick, solves/s."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, mmpc_loader
from oracle import synth, nlp
mm = mmpc_loader.load()
B, N, M, T = 8192, 30, 8, 10
par = nlp.WholeBodyParams(N=N)
dev = torch.device("cuda", 0)
for cid in [int(a) for a in sys.argv[1:]] or range(5, 13):
    d = synth.make_batch(B, N=N, M=M, config_id=cid, moving=True)
    glob = torch.from_numpy(d["traj_ref"]).to(dev)
    step = (glob[:, N] - glob[:, 0]) / N
    glob = glob[:, :1] + step[:, None, :] * torch.arange(51, dtype=torch.float64, device=dev)[None, :, None]
    fleet = mm.DeviceFleet(mm, np.clip(d["x_init"], par.xlim[0], par.xlim[1]), glob, d["obs"], d["obs_vel"], N=N, handles=1)
    out = []
    for name, fn in (("lock step", lambda: fleet.run_lockstep(T)), ("3 groups", lambda: fleet.run_groups(T, groups=3))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); el = time.perf_counter() - t0
        out.append("%s %.0f k solves/s (all converged: %s)" % (name, B * T / el / 1e3, bool(r["all_converged"])))
    it = r["iters"].cpu().numpy()
    print("seed %2d: %s | mean iterations per tick %.1f-%.1f, slowest robot per tick %s" % (cid, "; ".join(out), it.mean(0).min(), it.mean(0).max(), it.max(0).tolist()), flush=True)
    del fleet
