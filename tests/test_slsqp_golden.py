"""CPU: the committed second-source solutions (tests/golden/slsqp_solutions.npz, scipy SLSQP on the restated reference
NLP - oracle/gen_slsqp_golden.py) against the CPU oracle.  The GPU-side comparison is tests/test_gpu_certificates.py."""
import os

import numpy as np
import pytest

from oracle import nlp, coracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "slsqp_solutions.npz")


# Fixtures where the second solver did not end at the interior-point minimiser (the generator records whatever SLSQP
# returns; nothing is dropped).  For these the tests require agreement with the CPU oracle and the certificate instead.
#   other minimum: the NLP is non-convex (obstacles can be passed on either side); SLSQP's iterates end in another KKT point
#   stalled: SLSQP stops with "positive directional derivative" at a point that is not a KKT point (its certificate says so)
EXCEPTIONS = {
    "c5_1": "other minimum: SLSQP (slack-scaled run) ends at cost 899.37, the interior-point iterates at 1697.49 (its first, "
            "unscaled run stalled 1.6e-5 from that point); both are KKT points",
    "txy_0": "other minimum: SLSQP ends at cost 38.56, the interior-point iterates at 634.45; both satisfy the terminal equality "
             "and are KKT points (E0 2e-9)",
    "c1_tent": "kink of the max over the two planes: SLSQP (slack-scaled run) stops at cost 62.067, the interior-point iterates "
               "at the better 61.887; SLSQP's first run stalled at a non-KKT point",
}
# SLSQP's stopping accuracy: X to 1e-4; U to 5e-4 - the arm inputs carry no R weight (:14), only
# W = 0.1, so U is the least determined block of the minimiser (measured 1.6e-4 .. 2.9e-4 on c5_0, c5_3, c1_demo)
TOL_X, TOL_U, TOL_COST = 1e-4, 5e-4, 1e-6


def load_cases():
    z = np.load(GOLD)
    out = []
    for name in z["names"]:
        g = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}
        kind = "wholebody" if int(g["kind"]) == 0 else "base"
        par = nlp.WholeBodyParams(N=int(g["N"])) if kind == "wholebody" else nlp.BaseParams(N=int(g["N"]))
        par.Q, par.P = g["Q"], g["P"]
        par.terminal_xy_equality = bool(int(g["terminal_xy"]))
        out.append((str(name), par, g))
    return out


def test_fixture_inventory():
    names = [n for n, _, _ in load_cases()]
    assert len(names) >= 10 and len(names) - len(EXCEPTIONS) >= 10 and set(EXCEPTIONS) <= set(names)
    for prefix in ("c1_", "c2_", "c3_", "c5_", "txy_"):
        assert any(n.startswith(prefix) for n in names), prefix


@pytest.mark.parametrize("name,par,g", load_cases(), ids=[n for n, _, _ in load_cases()])
def test_oracle_reaches_the_slsqp_minimiser(name, par, g):
    """Stated cross-solver tolerance: |dX| <= 1e-4, |dU| <= 5e-4, cost 1e-6 relative; 13 of the 16 fixtures, the other
    three are listed in EXCEPTIONS with what happened."""
    hs = g["hs"] if len(g["hs"]) else None
    o = coracle.solve_batch(par, nlp.clip_x_init(par, g["x_init"])[None], g["traj_ref"][None], g["u_ref"][None], g["u_last"][None],
                            g["obs"][None], hs=hs, max_iter=2000)
    assert o["status"][0] == 0
    if name in EXCEPTIONS:
        prob = nlp.Problem(par, nlp.clip_x_init(par, g["x_init"]), g["traj_ref"], g["u_ref"], g["u_last"], g["obs"], hs)
        c = nlp.kkt_certificate_ipopt(prob, o["X"][0], o["U"][0], o["s"][0])
        assert c["E0"] <= 3e-8, (name, EXCEPTIONS[name], c)
        return
    assert abs(o["cost"][0] - float(g["cost"])) <= TOL_COST * abs(float(g["cost"]))
    assert np.abs(o["X"][0] - g["X"]).max() <= TOL_X and np.abs(o["U"][0] - g["U"]).max() <= TOL_U
