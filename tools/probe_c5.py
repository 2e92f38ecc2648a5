"""C5 ticks one by one: kernel time of each tick beside its mean / slowest instance (is a tick bound by its work or by its tail?)."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mmpc_loader; mm = mmpc_loader.load()
from oracle import synth
dev = torch.device("cuda", 0)
N, M, B, T = 30, 8, 8192, 10
d = synth.make_batch(B, N=N, M=M, config_id=int(os.environ.get("C5_SEED", 5)), moving=True)
ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=B, n_obstacles=M, obs_per_stage=True)
eng = ctrl._engine
f64 = dict(dtype=torch.float64, device=dev)
x = torch.from_numpy(np.clip(d["x_init"], ctrl.xlim[0], ctrl.xlim[1])).to(dev)
glob = torch.from_numpy(d["traj_ref"]).to(dev)
step = (glob[:, N] - glob[:, 0]) / N
glob = glob[:, :1] + step[:, None, :] * torch.arange(51, **f64)[None, :, None]
obs0 = torch.from_numpy(d["obs"]).to(dev); vel = torch.from_numpy(d["obs_vel"]).to(dev)
uref = torch.zeros((B, N, 5), **f64); ul = torch.zeros((B, N, 5), **f64)
xlo = torch.from_numpy(ctrl.xlim[0]).to(dev); xhi = torch.from_numpy(ctrl.xlim[1]).to(dev)
karr = torch.arange(N + 1, **f64)
out = None
SHIFTED = "--shifted" in sys.argv      # the engine's opt-in warm start (mmpc_set_warm_start) instead of the reference's protocol
ug = torch.zeros((B, N, 5), **f64); xg = torch.zeros((B, N + 1, 9), **f64)
def f_batch(xc, u):
    c, s = torch.cos(xc[:, 2]), torch.sin(xc[:, 2])
    return torch.stack([xc[:, 0] + 0.1 * xc[:, 3], xc[:, 1] + 0.1 * xc[:, 4], xc[:, 2] + 0.1 * xc[:, 5],
                        xc[:, 3] + 0.1 * (u[:, 0] * c - xc[:, 4] * xc[:, 5]), xc[:, 4] + 0.1 * (u[:, 0] * s + xc[:, 3] * xc[:, 5]),
                        xc[:, 5] + 0.1 * u[:, 1], xc[:, 6] + 0.1 * u[:, 2], xc[:, 7] + 0.1 * u[:, 3], xc[:, 8] + 0.1 * u[:, 4]], dim=1)
for t in range(T):
    dist = torch.linalg.norm(x[:, None, :2] - glob[:, :, :2], dim=2)
    idx = torch.clamp(torch.argmin(dist, dim=1)[:, None] + torch.arange(N + 1, device=dev)[None, :], max=50)
    loc = torch.gather(glob, 1, idx[:, :, None].expand(B, N + 1, 9)).contiguous()
    obs = obs0[:, None, :, :].repeat(1, N + 1, 1, 1)
    obs[..., :2] += vel[:, None, :, :] * ((t + karr) * 0.1)[None, :, None, None]
    obs = obs.contiguous()
    for rep in range(2):    # same tick twice: the second run is the timed one (first touches memory / LPT statistics)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        xgs = None
        if SHIFTED and t >= 1:
            ug[:, :-1] = ul[:, 1:]; ug[:, -1] = ul[:, -1]
            xg[:, 0] = torch.minimum(torch.maximum(x, xlo), xhi)
            for k in range(N):
                xg[:, k + 1] = f_batch(xg[:, k], ug[:, k])
            eng.set_warm_start(ug, 0.1); xgs = xg
            torch.cuda.synchronize(); e0.record()
        out = eng.solve_batch_device(x, loc, uref, ul, obs, x_guess=xgs, out=out)
        e1.record(); torch.cuda.synchronize()
    bad = torch.nonzero(out["status"] != 0).flatten().tolist()
    if bad:
        print("   not converged:", [(b, int(out["status"][b]), int(out["iters"][b])) for b in bad], flush=True)
        if "--dump" in sys.argv:
            for b in bad:
                np.savez(os.path.join(ROOT, "gpurun_out", "c5warmfail_t%d_b%d.npz" % (t, b)), x=x[b].cpu().numpy(), loc=loc[b].cpu().numpy(),
                         ul=ul[b].cpu().numpy(), obs=obs[b].cpu().numpy(), ug=ug[b].cpu().numpy(), xg=xg[b].cpu().numpy())
    if "--dump-slow" in sys.argv and SHIFTED and t >= 1:
        for b in torch.nonzero(out["iters"] > 120).flatten().tolist()[:6]:
            np.savez(os.path.join(ROOT, "gpurun_out", "c5warmslow_t%d_b%d.npz" % (t, b)), x=x[b].cpu().numpy(), loc=loc[b].cpu().numpy(),
                     ul=ul[b].cpu().numpy(), obs=obs[b].cpu().numpy(), ug=ug[b].cpu().numpy(), xg=xg[b].cpu().numpy(), iters=int(out["iters"][b]))
    it = out["iters"].double()
    top = torch.sort(out["iters"]).values[-4:].tolist()
    print("[%d per CU, %d B LDS] tick %d: %.2f ms  mean iters %.2f  top %s   work/slot at 3 per CU: %.0f iterations" %
          (eng.problems_per_cu, eng.lds_bytes, t, e0.elapsed_time(e1), it.mean().item(), top, it.sum().item() / 768), flush=True)
    if "--dump" in sys.argv:      # inputs of very slow instances, for offline study with the numpy restatement
        for b in torch.nonzero(out["iters"] > 400).flatten().tolist():
            np.savez(os.path.join(ROOT, "gpurun_out", "c5tail_t%d_b%d.npz" % (t, b)), x=x[b].cpu().numpy(), loc=loc[b].cpu().numpy(),
                     ul=ul[b].cpu().numpy(), obs=obs[b].cpu().numpy(), iters=int(out["iters"][b]))
    if "--iters" in sys.argv:
        np.save(os.path.join(ROOT, "gpurun_out", "c5_iters_t%d.npy" % t), out["iters"].cpu().numpy())
        # the a-priori difficulty key of this tick's data (as mmpc_difficulty_key) and a few other features, for offline schedule studies
        dd = torch.linalg.norm(loc[:, :, None, :2] - obs[..., :2], dim=3)                     # [B, N+1, M]
        pen = (obs[..., 2] + 0.4) - dd
        xd = torch.linalg.norm(x[:, None, None, :2] - obs[:, :1, :, :2], dim=3)[:, 0]        # start position to the discs at stage 0
        np.savez(os.path.join(ROOT, "gpurun_out", "c5_feat_t%d.npz" % t), key=pen.amax(dim=(1, 2)).cpu().numpy(), key_first=pen[:, :8].amax(dim=(1, 2)).cpu().numpy(),
                 ncut=(pen.amax(dim=1) > 0).sum(dim=1).cpu().numpy(), start_clear=(xd - obs[:, 0, :, 2] - 0.4).amin(dim=1).cpu().numpy(),
                 speed=torch.linalg.norm(x[:, 3:5], dim=1).cpu().numpy())
    ul = out["U"].clone(); u0 = out["U"][:, 0]
    xc = torch.minimum(torch.maximum(x, xlo), xhi)
    c, s = torch.cos(xc[:, 2]), torch.sin(xc[:, 2])
    x = torch.stack([xc[:, 0] + 0.1 * xc[:, 3], xc[:, 1] + 0.1 * xc[:, 4], xc[:, 2] + 0.1 * xc[:, 5],
                     xc[:, 3] + 0.1 * (u0[:, 0] * c - xc[:, 4] * xc[:, 5]), xc[:, 4] + 0.1 * (u0[:, 0] * s + xc[:, 3] * xc[:, 5]),
                     xc[:, 5] + 0.1 * u0[:, 1], xc[:, 6] + 0.1 * u0[:, 2], xc[:, 7] + 0.1 * u0[:, 3], xc[:, 8] + 0.1 * u0[:, 4]], dim=1)
