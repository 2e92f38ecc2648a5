"""CPU (host build of the kernel source): a solve cut into launches of at most `budget` iterations - suspended, state
written to the save area, resumed - runs exactly the iterations of the uninterrupted solve: bitwise equal outputs."""
import numpy as np
import pytest

import emu_helper
from oracle import nlp, synth


@pytest.mark.parametrize("kind,N,M,cid,budget", [("wholebody", 20, 5, 3, 7), ("base", 15, 3, 2, 3)])
def test_budgeted_launches_equal_the_uninterrupted_solve(kind, N, M, cid, budget):
    B = 6
    d = synth.make_batch(B, N=N, M=M, kind=kind, config_id=cid)
    par = nlp.WholeBodyParams(N=N) if kind == "wholebody" else nlp.BaseParams(N=N)
    xi = nlp.clip_x_init(par, d["x_init"]) if kind == "wholebody" else d["x_init"]
    ul = np.zeros((B, N, par.nu))
    ref = emu_helper.solve_batch(par, xi, d["traj_ref"], d["u_ref"], ul, d["obs"], fast=True, max_iter=2000)
    cut, launches = emu_helper.solve_fast_budgeted(par, xi, d["traj_ref"], d["u_ref"], ul, d["obs"], budget, max_iter=2000)
    assert (ref["status"] == 0).all() and (cut["status"] == 0).all()
    assert launches == int(np.ceil(ref["iters"].max() / budget)) or launches == int(np.ceil((ref["iters"].max() + 1) / budget))
    for k in ("X", "U", "s", "iters", "cost", "err"):
        assert np.array_equal(ref[k], cut[k]), k
