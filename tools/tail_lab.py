"""CPU lab of the solver (round 4): the C4 seed sweep and the C5 closed loop through the C oracle (oracle/mmpc_oracle.c, OpenMP), with the
oracle's experiment knobs (mmpc_oracle_set_lab: --lab i=v,...).  Prints iteration statistics, backward passes per solve, the slowest
instance per batch / tick.  Numbers behind DESIGN.md section 3 and profiles/NOTES_r04.md.  Generated batches are cached under
$MMPC_LAB_CACHE (default /tmp/mmpc_lab_cache).
  python tools/tail_lab.py c4 [--seeds 3,4] [--lab 1=0,2=0]      python tools/tail_lab.py c5 [--seeds 5,6] [--save-hard 150]"""
import sys, os, time, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
CACHE = os.environ.get("MMPC_LAB_CACHE", "/tmp/mmpc_lab_cache")
from oracle import coracle, nlp, synth
L = coracle.lib()
L.mmpc_oracle_set_lab.argtypes = [C.c_int, C.c_double]
L.mmpc_oracle_get_lab.restype = C.c_double
L.mmpc_oracle_get_lab.argtypes = [C.c_int]
NT = 8

def _batch(kind, cid):
    os.makedirs(CACHE, exist_ok=True)
    fn = os.path.join(CACHE, "%s_%d.npz" % (kind, cid))
    if not os.path.exists(fn):
        d = synth.make_batch(8192, config_id=cid) if kind == "c4" else synth.make_batch(8192, N=30, M=8, config_id=cid, moving=True)
        np.savez(fn, **d)
    return np.load(fn)


def setlab(**kw):
    for k, v in kw.items():
        L.mmpc_oracle_set_lab(int(k[1:]), float(v))

def c4(seeds=range(3, 13), B=8192, verbose=True, max_iter=2000):
    par = nlp.WholeBodyParams()
    tot = []
    for cid in seeds:
        d = _batch("c4", cid)
        x = np.clip(d["x_init"][:B], par.xlim[0], par.xlim[1])
        f0 = L.mmpc_oracle_get_lab(100); t0 = time.time()
        o = coracle.solve_batch(par, x, d["traj_ref"][:B], d["u_ref"][:B], np.zeros((B, 20, 5)), d["obs"][:B], nthreads=NT, max_iter=max_iter)
        nf = L.mmpc_oracle_get_lab(100) - f0
        it = o["iters"]
        top = np.sort(it)[-3:]
        tot.append((it.mean(), it.max(), (o["status"] != 0).sum(), nf / B, o["cost"].mean()))
        if verbose:
            print("c4 seed %2d: fail %d  iters mean %.2f p99 %.0f top3 %s  fact/solve %.2f  cost mean %.4f  (%.1fs)" % (cid, (o["status"] != 0).sum(), it.mean(), np.percentile(it, 99), top, nf / B, o["cost"].mean(), time.time() - t0), flush=True)
    tot = np.array(tot)
    print("C4 SUMMARY: mean iters %.3f  max-per-seed %s  worst %d  fails %d  fact/solve %.2f cost %.4f" % (tot[:, 0].mean(), tot[:, 1].astype(int).tolist(), tot[:, 1].max(), tot[:, 2].sum(), tot[:, 3].mean(), tot[:,4].mean()), flush=True)
    return tot

def plant(x, u, par, dt=0.1):
    xc = np.clip(x, par.xlim[0], par.xlim[1])
    c, s = np.cos(xc[:, 2]), np.sin(xc[:, 2])
    return np.stack([xc[:, 0] + dt * xc[:, 3], xc[:, 1] + dt * xc[:, 4], xc[:, 2] + dt * xc[:, 5],
                     xc[:, 3] + dt * (u[:, 0] * c - xc[:, 4] * xc[:, 5]), xc[:, 4] + dt * (u[:, 0] * s + xc[:, 3] * xc[:, 5]),
                     xc[:, 5] + dt * u[:, 1], xc[:, 6] + dt * u[:, 2], xc[:, 7] + dt * u[:, 3], xc[:, 8] + dt * u[:, 4]], axis=1)

def c5(seeds=range(5, 13), B=8192, T=10, verbose=True, save_hard=None, max_iter=2000):
    N, M = 30, 8
    par = nlp.WholeBodyParams(N=N)
    allmax = []; means = []; fails = 0
    for cid in seeds:
        d = _batch("c5", cid)
        glob = d["traj_ref"][:B]
        step = (glob[:, N] - glob[:, 0]) / N
        glob = glob[:, :1] + step[:, None, :] * np.arange(51.0)[None, :, None]
        x = np.clip(d["x_init"][:B], par.xlim[0], par.xlim[1]); ul = np.zeros((B, N, 5)); uref = np.zeros((B, N, 5))
        obs0 = d["obs"][:B]; vel = d["obs_vel"][:B]
        mx = []; t0 = time.time()
        for t in range(T):
            dd = np.linalg.norm(x[:, None, :2] - glob[:, :, :2], axis=2)
            start = np.argmin(dd, axis=1)
            idx = np.minimum(start[:, None] + np.arange(N + 1)[None, :], 50)
            loc = np.take_along_axis(glob, idx[:, :, None], axis=1)
            obs = np.repeat(obs0[:, None], N + 1, axis=1).copy()
            tk = (t + np.arange(N + 1.0)) * 0.1
            obs[..., :2] += vel[:, None, :, :] * tk[None, :, None, None]
            xin = np.clip(x, par.xlim[0], par.xlim[1])
            o = coracle.solve_batch(par, xin, loc, uref, ul, obs, nthreads=NT, max_iter=max_iter)
            it = o["iters"]; mx.append(int(it.max())); means.append(it.mean()); fails += int((o["status"] != 0).sum())
            if save_hard is not None:
                for b in np.nonzero(it > save_hard)[0]:
                    np.savez(os.path.join(CACHE, "hard", "c5_s%d_t%d_b%d.npz" % (cid, t, b)), x=xin[b], loc=loc[b], ul=ul[b], obs=obs[b], iters=it[b])
            conv = o["status"] == 0
            ul = np.where(conv[:, None, None], o["U"], ul)
            x = plant(x, o["U"][:, 0], par)
        allmax.append(mx)
        if verbose:
            print("c5 seed %2d: slowest per tick %s  (%.0fs)" % (cid, mx, time.time() - t0), flush=True)
    am = np.array(allmax)
    print("C5 SUMMARY: mean iters %.2f  worst %d  ticks>400: %d of %d  fails %d" % (np.mean(means), am.max(), (am > 400).sum(), am.size, fails), flush=True)
    return am

if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("what"); ap.add_argument("--lab", default=""); ap.add_argument("--seeds", default=""); ap.add_argument("--B", type=int, default=8192)
    ap.add_argument("--save-hard", type=int, default=None)
    a = ap.parse_args()
    for kv in a.lab.split(","):
        if kv:
            k, v = kv.split("="); L.mmpc_oracle_set_lab(int(k), float(v))
    seeds = [int(s) for s in a.seeds.split(",")] if a.seeds else None
    if a.what == "c4": c4(seeds or range(3, 13), B=a.B)
    elif a.what == "c5":
        os.makedirs(os.path.join(CACHE, "hard"), exist_ok=True)
        c5(seeds or range(5, 13), B=a.B, save_hard=a.save_hard)
