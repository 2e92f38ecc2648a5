"""Runs oracle.nlp.kkt_certificate_ipopt for many (problem, solution) pairs on the host cores: plain worker
subprocesses (the parent of a GPU test has the HIP runtime loaded and must not be forked), one BLAS thread each."""
import os
import pickle
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(path_in, path_out):
    sys.path.insert(0, ROOT)
    from oracle import nlp
    with open(path_in, "rb") as f:
        items = pickle.load(f)
    out = []
    for prob, X, U, s in items:
        out.append(nlp.kkt_certificate_ipopt(prob, X, U, s))
        with open(path_out + ".n", "w") as f:
            f.write(str(len(out)))
    with open(path_out, "wb") as f:
        pickle.dump(out, f)


def certify(items, procs=None, label="certificates", timeout=900):
    """items: list of (nlp.Problem, X, U, s) -> list of certificate dicts (same order).  Prints a progress line every
    ~20 s (a long silent run on the GPU box is taken to be hung)."""
    items = list(items)
    if not items:
        return []
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(procs or 16, avail, len(items)))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONPATH=ROOT)
    t0 = last = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        jobs = []
        for w in range(procs):
            pin, pout = os.path.join(tmp, "in%d.pkl" % w), os.path.join(tmp, "out%d.pkl" % w)
            with open(pin, "wb") as f:
                pickle.dump(items[w::procs], f)
            jobs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), pin, pout], env=env), pout))
        while any(p.poll() is None for p, _ in jobs):
            time.sleep(0.5)
            if time.time() - t0 > timeout:
                for p, _ in jobs:
                    p.kill()
                raise RuntimeError("%s: workers exceeded %d s" % (label, timeout))
            if time.time() - last > 20:
                last = time.time()
                done = sum(int(open(po + ".n").read() or 0) for _, po in jobs if os.path.exists(po + ".n"))
                print("  %s: %d / %d after %.0f s (%d workers)" % (label, done, len(items), last - t0, procs), flush=True)
        out = [None] * len(items)
        for w, (p, pout) in enumerate(jobs):
            if p.returncode != 0:
                raise RuntimeError("%s: worker %d failed (%d)" % (label, w, p.returncode))
            with open(pout, "rb") as f:
                out[w::procs] = pickle.load(f)
    return out


if __name__ == "__main__":
    _worker(sys.argv[1], sys.argv[2])
