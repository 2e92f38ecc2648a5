import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, mmpc_loader
from oracle import synth, nlp
mm = mmpc_loader.load()
B, N, M, T = 8192, 30, 8, 10
d = synth.make_batch(B, N=N, M=M, config_id=5, moving=True)
par = nlp.WholeBodyParams(N=N)
dev = torch.device("cuda", 0)
glob = torch.from_numpy(d["traj_ref"]).to(dev)
step = (glob[:, N] - glob[:, 0]) / N
glob = glob[:, :1] + step[:, None, :] * torch.arange(51, dtype=torch.float64, device=dev)[None, :, None]
fleet = mm.DeviceFleet(mm, np.clip(d["x_init"], par.xlim[0], par.xlim[1]), glob, d["obs"], d["obs_vel"], N=N)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return B * T * reps / (time.perf_counter() - t0), r
v, a = timeit(lambda: fleet.run_lockstep(T)); print("lockstep %.0f" % v, flush=True)
for G in (2, 3, 4, 6, 8):
    for pr in (True, False):
        v, r = timeit(lambda: fleet.run_groups(T, groups=G, priority=pr))
        print("groups %d priority %s: %.0f solves/s  equal %s" % (G, pr, v, bool(torch.equal(a["u0"], r["u0"]))), flush=True)
