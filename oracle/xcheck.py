"""Independent second-source solver for the NLP of oracle/nlp.py (scipy SLSQP).

TEST INFRASTRUCTURE ONLY.  Correctness cross-check, never timed, never shipped.
SLSQP is an active-set SQP method - a different algorithm family from both IPOPT
(the reference's solver, controllers/mpc_wholebody_qref.py:285) and the build's own
interior-point solver - so agreement of the minimisers is independent evidence.
"""
import numpy as np
from scipy.optimize import minimize
from . import nlp


def solve_slsqp(prob: nlp.Problem, U0=None, maxiter=400, ftol=1e-13, slack_scale=1.0):
    """slack_scale: the optimiser works on sigma = slack_scale * s (a change of variables of the same NLP; sqrt(S)
    conditions the S s^2 term for SLSQP's QP subproblems)."""
    p = prob.par
    N = p.N
    ns = N + 1
    X0 = np.tile(prob.x_init, (N + 1, 1))
    U0 = np.zeros((N, p.nu)) if U0 is None else U0
    w0 = nlp.pack(X0, U0, np.zeros(N + 1))

    def un(w):
        X, U, s = nlp.unpack(p, w)
        return X, U, s / slack_scale

    def sg(v):
        v = v.copy(); v[-ns:] /= slack_scale
        return v

    def sj(J):
        J = J.copy(); J[:, -ns:] /= slack_scale
        return J

    def f(w):
        return nlp.cost(prob, *un(w))

    def g(w):
        return sg(nlp.pack(*nlp.cost_grad(prob, *un(w))))

    cons = [
        dict(type="eq", fun=lambda w: nlp.eq_rows(prob, *un(w)[:2]), jac=lambda w: nlp.eq_jac(prob, *un(w)[:2])),
        dict(type="ineq", fun=lambda w: -nlp.ineq_rows(prob, *un(w)), jac=lambda w: -sj(nlp.ineq_jac(prob, *un(w)))),
    ]
    r = minimize(f, w0, jac=g, constraints=cons, method="SLSQP",
                 options=dict(maxiter=maxiter, ftol=ftol))
    X, U, s = un(r.x)
    return dict(X=X, U=U, s=s, cost=r.fun, iters=r.nit, success=r.success, message=r.message)
