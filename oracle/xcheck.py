"""Independent second-source solver for the NLP of oracle/nlp.py (scipy SLSQP).

TEST INFRASTRUCTURE ONLY.  Correctness cross-check, never timed, never shipped.
SLSQP is an active-set SQP method - a different algorithm family from both IPOPT
(the reference's solver, controllers/mpc_wholebody_qref.py:285) and the build's own
interior-point solver - so agreement of the minimisers is independent evidence.
"""
import numpy as np
from scipy.optimize import minimize
from . import nlp


def solve_slsqp(prob: nlp.Problem, U0=None, maxiter=400, ftol=1e-13):
    p = prob.par
    N = p.N
    X0 = np.tile(prob.x_init, (N + 1, 1))
    U0 = np.zeros((N, p.nu)) if U0 is None else U0
    w0 = nlp.pack(X0, U0, np.zeros(N + 1))

    def f(w):
        X, U, s = nlp.unpack(p, w)
        return nlp.cost(prob, X, U, s)

    def g(w):
        X, U, s = nlp.unpack(p, w)
        return nlp.pack(*nlp.cost_grad(prob, X, U, s))

    cons = [
        dict(type="eq", fun=lambda w: nlp.eq_rows(prob, *nlp.unpack(p, w)[:2]),
             jac=lambda w: nlp.eq_jac(prob, *nlp.unpack(p, w)[:2])),
        dict(type="ineq", fun=lambda w: -nlp.ineq_rows(prob, *nlp.unpack(p, w)),
             jac=lambda w: -nlp.ineq_jac(prob, *nlp.unpack(p, w))),
    ]
    r = minimize(f, w0, jac=g, constraints=cons, method="SLSQP",
                 options=dict(maxiter=maxiter, ftol=ftol))
    X, U, s = nlp.unpack(p, r.x)
    return dict(X=X, U=U, s=s, cost=r.fun, iters=r.nit, success=r.success, message=r.message)
