"""Reference implementation (numpy, one instance at a time) of the solver algorithm that
the HIP kernels and the C port (oracle/mmpc_oracle.c) run.

TEST INFRASTRUCTURE ONLY (see oracle/nlp.py header).  Slow and explicit on purpose.

The reference hands the NLP of controllers/mpc_wholebody_qref.py:142-285 to IPOPT
(primal-dual interior point, exact Hessian, :280-285).  This file states the build's
own solver for the same NLP:

  * primal-dual interior point on the inequality rows, one slack t_i>0 and one
    multiplier z_i>0 per row (h_i(w) + t_i = 0), monotone barrier schedule
    (mu <- max(tol/10, min(0.2 mu, mu^1.5)), subproblem tolerance 10 mu) - the
    Fiacco-McCormick schedule IPOPT uses by default;
  * exact Lagrangian Hessian (cost + dynamics curvature + row curvature), the row
    blocks condensed stage-wise; the per-stage scalar slack s_k is removed by a
    rank-one Schur complement (quirk Q1 handled: terminal self-collision rows are
    bound to s_{N-1});
  * the Newton system is the block-tridiagonal KKT system of the multiple-shooting
    QP; it is factorised stage by stage (Riccati recursion = block LDL^T of the KKT
    matrix), with on-the-fly diagonal regularisation of a stage's input Hessian
    when its Cholesky pivot is not positive;
  * fraction-to-boundary rule + backtracking line search on an l1 merit function.
"""
from __future__ import annotations

import numpy as np
from . import nlp


class Options:
    tol = 1e-8
    mu_init = 1.0
    max_iter = 200
    kappa_eps = 10.0
    kappa_mu = 0.2
    theta_mu = 1.5
    tau_min = 0.99
    t_init_min = 1e-2
    eta = 1e-4
    max_ls = 20
    reg0 = 1e-6
    exact_hessian = True
    curv_dyn = True
    curv_circ = True
    curv_self = False
    curv_hs = True        # exact curvature of the half-space rows (n . FK point): their multipliers are large (S = 1e5)
    piv_frac = 0.0
    nu_lam = 0.0
    filter = True
    mid_fallback = True     # (round-3 ladder, only with inertia = False) exact -> exact without dynamics curvature -> Gauss-Newton
    inertia = True          # IPOPT's inertia correction (Waechter & Biegler 2006, Algorithm IC): exact Hessian + delta_w I, delta_w from a
                            # geometric sequence (delta0, x kappa_first while no correction has succeeded yet, x kappa_up after, the next
                            # iteration starts from kappa_dn times the last successful value) until every pivot of the recursion is positive
    kappa_first = 4.0; kappa_up = 4.0; kappa_dn = 1.0 / 3.0
    ic_skip0 = 0.1          # an iteration whose predecessor needed a correction above this starts from the corrected matrix at once (IPOPT always
                            # tries delta_w = 0 first: here that attempt failed in most iterations of a run of corrected ones - a wasted pass)
    soc_max = 2             # second-order corrections per iteration (IPOPT: max_soc = 4), tried when the first trial step is rejected
                            # without reducing the infeasibility and theta(x_k) <= theta_min (the regime of the switching condition)
    inertia_streak = 0      # >0 (experiment, off: it trades the 600-800-iteration cases for others that take 1300+): inertia correction (exact Hessian + delta I) instead of the Gauss-Newton fallback once the
                            # exact Hessian has failed in this many consecutive iterations (slow linear convergence near saddles)
    slack_reset = False
    delta0 = 1.0          # (IPOPT: 1e-4 and x100; the corrections this NLP needs are 1 ... 100)
    delta_min = 1e-20
    delta_max = 1e10
    bound_push = 1e-2
    kappa_sigma = 1e10
    # proximal term for crawling iterations: after two consecutive steps with alpha < prox_lo the Hessian gets + prox I on
    # (x, u) (prox0, then x prox_up per further small step); it is divided by prox_dn after a step with alpha > prox_hi
    prox = True; prox_on_ap = True; prox0 = 100.0; prox_up = 4.0; prox_dn = 4.0; prox_lo = 0.05; prox_hi = 0.5; prox_need = 2; prox_max = 1e4
    rho_eq = 1e4          # augmentation weight of the terminal-xy equality inside the factorisation


def _rows_for_stage(prob, k):
    """Static description of the inequality rows of stage k in the solver's own order.

    u rows (k<N): merged box  max(ulim_lo, u_last+dulim_lo) <= u <= min(ulim_hi, u_last+dulim_hi)
    (mpc_wholebody_qref.py:203,205 - both are simple bounds on the same variable);
    x rows (k>=1; x_0 is data, :244); circle rows (:208-209,:248-249); self rows (:219-222,:261-265).
    Returns list of tuples (kind, index, sign/bound...).
    """
    p = prob.par
    rows = []
    if k < p.N:
        lo = np.maximum(p.ulim[0], prob.u_last[k] + p.dulim[0])
        hi = np.minimum(p.ulim[1], prob.u_last[k] + p.dulim[1])
        for j in range(p.nu):
            if np.isfinite(lo[j]):
                rows.append(("ulo", j, lo[j]))
            if np.isfinite(hi[j]):
                rows.append(("uhi", j, hi[j]))
    if k >= 1:
        for j in range(p.nx):
            if np.isfinite(p.xlim[0, j]):
                rows.append(("xlo", j, p.xlim[0, j]))
            if np.isfinite(p.xlim[1, j]):
                rows.append(("xhi", j, p.xlim[1, j]))
    for m, o in enumerate(prob.obs_at(k)):
        rows.append(("circ", m, o))
    if nlp.has_arm_rows(p):
        for i in range(4):
            rows.append(("self", i, None))
        if prob.hs is not None and len(prob.hs):
            for i in range(6):
                rows.append(("hs", i, None))
    return rows


def _eval_row(prob, k, row, X, U, s, order):
    """h (<=0 form), and sparse derivative pieces: (h, jx, ju, js_index, Hxx)."""
    p = prob.par
    kind, j, b = row
    nx, nu = p.nx, p.nu
    if kind == "ulo":
        ju = np.zeros(nu); ju[j] = -1
        return b - U[k, j], None, ju, None, None
    if kind == "uhi":
        ju = np.zeros(nu); ju[j] = 1
        return U[k, j] - b, None, ju, None, None
    if kind == "xlo":
        jx = np.zeros(nx); jx[j] = -1
        return b - X[k, j], jx, None, None, None
    if kind == "xhi":
        jx = np.zeros(nx); jx[j] = 1
        return X[k, j] - b, jx, None, None, None
    if kind == "circ":
        if order == 0:
            return nlp.circle_row(X[k], b, nx, 0) - s[k], None, None, k, None
        g, jx, H = nlp.circle_row(X[k], b, nx, 2)
        return g - s[k], jx, None, k, H
    if kind == "self":
        ks = nlp.slack_index(p, k)
        if order == 0:
            return nlp.selfcol_row(X[k], j, 0) - s[ks], None, None, ks, None
        h, jx, H = nlp.selfcol_row(X[k], j, 2)
        return h - s[ks], jx, None, ks, H
    if kind == "hs":
        if order == 0:
            return nlp.halfspace_row(X[k], j, prob.hs, 0) - s[k], None, None, k, None
        h, jx, H = nlp.halfspace_row(X[k], j, prob.hs, 2)
        return h - s[k], jx, None, k, H
    raise ValueError(kind)


class State:
    pass


def _bound_push(v, lo, hi, kap):
    """clip v into [lo + kap, hi - kap]; a box narrower than 2 kap collapses to its midpoint"""
    lo2, hi2 = lo + kap, hi - kap
    with np.errstate(invalid="ignore"):
        mid = 0.5 * (lo + hi)            # (nan for a free variable: never selected)
    narrow = lo2 > hi2
    lo2 = np.where(narrow, mid, lo2); hi2 = np.where(narrow, mid, hi2)
    return np.minimum(np.maximum(v, lo2), hi2)


def _eval_all(prob, rows, X, U, s, order):
    out = []
    for k in range(prob.par.N + 1):
        out.append([_eval_row(prob, k, r, X, U, s, order) for r in rows[k]])
    return out


def _defects(prob, X, U):
    p = prob.par
    return np.array([nlp.f_dyn(p.kind, X[k], U[k], p.dt) - X[k + 1] for k in range(p.N)])


def _merit_parts(prob, rows, X, U, s, t, mu):
    """returns (barrier objective, l1 infeasibility)"""
    p = prob.par
    f = nlp.cost(prob, X, U, s)
    c = _defects(prob, X, U)
    th = np.abs(c).sum() + np.abs(X[0] - prob.x_init).sum()
    if p.terminal_xy_equality:
        th += np.abs(X[p.N, :2] - prob.traj_ref[p.N, :2]).sum()
    bar = 0.0
    for k in range(p.N + 1):
        for i, r in enumerate(rows[k]):
            h = _eval_row(prob, k, r, X, U, s, 0)[0]
            th += abs(h + t[k][i])
            bar -= mu * np.log(t[k][i])
    return f + bar, th


def solve(prob: nlp.Problem, U0=None, X0=None, opt: Options = None, verbose=False, history=None):
    """Solve one instance.  Initial point per the reference protocol
    (mpc_wholebody_qref.py:302-304): X = tile(x_init), U = u_latest, s = 0.
    Returns dict(X,U,s,status,iters,cost,kkt)."""
    opt = opt or Options()
    p = prob.par
    N, nx, nu = p.N, p.nx, p.nu
    X = np.tile(prob.x_init, (N + 1, 1)) if X0 is None else np.array(X0, float)
    X[0] = prob.x_init
    U = np.zeros((N, nu)) if U0 is None else np.array(U0, float)
    s = np.zeros(N + 1)
    lam = np.zeros((N + 1, nx))      # lam[k+1]: multiplier of x_{k+1} - f(x_k,u_k) = 0
    # bound push (IPOPT's treatment of the initial point, bound_push = 1e-2): a variable with a finite simple bound
    # starts at least kappa inside it.  The box rows are linear, so their slack then equals the distance to the bound
    # for the whole solve (t = v - lo), which is what lets the GPU kernel keep no slack state for them.
    for k in range(N + 1):
        if k < N:
            lo = np.maximum(p.ulim[0], prob.u_last[k] + p.dulim[0])
            hi = np.minimum(p.ulim[1], prob.u_last[k] + p.dulim[1])
            U[k] = _bound_push(U[k], lo, hi, opt.bound_push)
        if k >= 1:
            X[k] = _bound_push(X[k], p.xlim[0], p.xlim[1], opt.bound_push)
    rows = [_rows_for_stage(prob, k) for k in range(N + 1)]
    mu = opt.mu_init
    # slacks / multipliers
    ev = _eval_all(prob, rows, X, U, s, 0)
    t = [np.array([max(-e[0], opt.t_init_min) for e in ev[k]]) for k in range(N + 1)]
    z = [mu / t[k] for k in range(N + 1)]
    nrows = sum(len(r) for r in rows)
    Q2 = p.Q + p.Q.T
    P2 = p.P + p.P.T
    RW2 = (p.R + p.R.T) + (p.W + p.W.T)
    status = 1
    nu_merit = 1.0
    delta_last = 0.0; delta_used = 0.0; delta_prev_it = 0.0
    nu_eq = np.zeros(2)
    prox_cur, nsmall, nstreak = 0.0, 0, 0
    filt = None
    nfail = 0
    it = 0
    for it in range(opt.max_iter + 1):
        # ---------------- multiplier safeguard (IPOPT eq. 16, kappa_Sigma = 1e10) --------
        # keeps z_i within [mu / (kappa t_i), kappa mu / t_i]: full dual steps next to tiny primal ones cannot run away
        for k in range(N + 1):
            z[k] = np.minimum(np.maximum(z[k], mu / (opt.kappa_sigma * t[k])), opt.kappa_sigma * mu / t[k])
        # ---------------- evaluation ------------------------------------
        ev = _eval_all(prob, rows, X, U, s, 2)
        gX, gU, gs = nlp.cost_grad(prob, X, U, s)
        c = _defects(prob, X, U)
        AB = [nlp.f_jac(p.kind, X[k], U[k], p.dt) for k in range(N)]
        # ---------------- KKT error at current point ---------------------
        rdx = gX.copy(); rdu = gU.copy(); rds = gs.copy()
        for k in range(N + 1):
            for i, (h, jx, ju, ks, H) in enumerate(ev[k]):
                if jx is not None: rdx[k] += jx * z[k][i]
                if ju is not None: rdu[k] += ju * z[k][i]
                if ks is not None: rds[ks] -= z[k][i]
        # equality multipliers: L = f + sum lam[k+1]^T (x_{k+1} - f(x_k,u_k)) + lam0^T(x0 - xinit)
        for k in range(N):
            A, B = AB[k]
            rdx[k + 1] += lam[k + 1]
            rdx[k] -= A.T @ lam[k + 1]
            rdu[k] -= B.T @ lam[k + 1]
        rdx[0] = 0.0   # x_0 is fixed data; its multiplier absorbs the residual
        ceq = np.zeros(2)
        if p.terminal_xy_equality:
            rdx[N][:2] += nu_eq
            ceq = X[N, :2] - prob.traj_ref[N, :2]
        if opt.slack_reset:
            # slack reset: a row that is more satisfied than its slack says (-h > t) gets t := -h; this removes that
            # row's infeasibility for free and moves the slack away from the boundary (cf. Nocedal & Wright 19.3)
            for k in range(N + 1):
                hv = np.array([e[0] for e in ev[k]])
                t[k] = np.maximum(t[k], -hv)
        rh = [np.array([e[0] for e in ev[k]]) + t[k] for k in range(N + 1)]
        err_d = max(np.abs(rdx).max(), np.abs(rdu).max(), np.abs(rds).max())
        err_p = max(np.abs(c).max() if N else 0.0, max(np.abs(r).max() for r in rh), np.abs(ceq).max())
        comp0 = max((t[k] * z[k]).max() for k in range(N + 1))
        compmu = max(np.abs(t[k] * z[k] - mu).max() for k in range(N + 1))
        zsum = sum(z[k].sum() for k in range(N + 1)) + np.abs(lam).sum() + np.abs(nu_eq).sum()
        sd = max(100.0, zsum / (nrows + lam.size)) / 100.0
        # IPOPT's scaling of the complementarity (Waechter & Biegler 2006, eq. 5-6): s_c from the row multipliers alone
        sc = max(100.0, sum(z[k].sum() for k in range(N + 1)) / max(nrows, 1)) / 100.0
        E0 = max(err_d / sd, err_p, comp0 / sc)
        Emu = max(err_d / sd, err_p, compmu / sc)
        if history is not None:
            history.append(dict(it=it, mu=mu, E0=E0, Emu=Emu, err_d=err_d, err_p=err_p,
                                cost=nlp.cost(prob, X, U, s), X=X.copy()))
        if verbose:
            print(f"it {it:3d} mu {mu:8.2e} E0 {E0:9.3e} Emu {Emu:9.3e} d {err_d:9.3e} p {err_p:9.3e} "
                  f"f {nlp.cost(prob, X, U, s):12.6f}")
        if E0 <= opt.tol:
            status = 0
            break
        if it == opt.max_iter:
            break
        changed = False
        while Emu <= opt.kappa_eps * mu and mu > opt.tol / 10:
            mu = max(opt.tol / 10, min(opt.kappa_mu * mu, mu ** opt.theta_mu))
            compmu = max(np.abs(t[k] * z[k] - mu).max() for k in range(N + 1))
            Emu = max(err_d / sd, err_p, compmu / sc)
            changed = True
        if changed:
            nu_merit = 1.0
            filt = None
        # ---------------- stage QP assembly + Riccati ----------------------
        def factor(use_exact, delta=0.0, dyn_curv=True):
            Hxx = []; Hux = []; Huu = []; qx = []; qu = []
            hss = 2 * p.S * np.ones(N + 1)
            gss = gs.copy()
            vx = np.zeros((N + 1, nx))        # sum_i w_i J_i,x for rows bound to s_k from stage k
            vxN = np.zeros(nx)                # Q1: stage-N self rows bound to s_{N-1}
            for k in range(N + 1):
                _, _, Hgn, Hcv = nlp.state_cost(p, k, X[k], prob.traj_ref[k], 2)
                H = Hgn + (Hcv if use_exact else 0.0) + delta * np.eye(nx)
                Hu = RW2 + delta * np.eye(nu) if k < N else None
                Hc = np.zeros((nu, nx)) if k < N else None
                g = gX[k].copy()
                gu = gU[k].copy() if k < N else None
                if k < N and use_exact and opt.curv_dyn and dyn_curv:
                    fxx, fux, fuu = nlp.f_hess_contract(p.kind, X[k], U[k], p.dt, lam[k + 1])
                    H -= fxx; Hc -= fux; Hu -= fuu
                for i, (h, jx, ju, ks, Hr) in enumerate(ev[k]):
                    w = z[k][i] / t[k][i]
                    zh = mu / t[k][i] + w * rh[k][i]
                    if jx is not None:
                        H += w * np.outer(jx, jx)
                        g += jx * zh
                    if ju is not None:
                        Hu += w * np.outer(ju, ju)
                        gu += ju * zh
                    if Hr is not None and use_exact:
                        if (rows[k][i][0] == "circ" and opt.curv_circ) or (rows[k][i][0] == "self" and opt.curv_self) or \
                                (rows[k][i][0] == "hs" and opt.curv_hs):
                            H += z[k][i] * Hr
                    if ks is not None:
                        hss[ks] += w
                        gss[ks] -= zh
                        if ks == k:
                            vx[k] += w * jx
                        else:
                            vxN += w * jx
                Hxx.append(H); Hux.append(Hc); Huu.append(Hu); qx.append(g); qu.append(gu)
            # Schur complement of s_k  (cross term H_xs = -v)
            for k in range(N + 1):
                if k == N - 1:
                    A, B = AB[k]
                    a = vx[k] + A.T @ vxN
                    b = B.T @ vxN
                    gam = gss[k] - vxN @ c[k]
                    Hxx[k] -= np.outer(a, a) / hss[k]
                    Hux[k] -= np.outer(b, a) / hss[k]
                    Huu[k] -= np.outer(b, b) / hss[k]
                    qx[k] += a * gam / hss[k]
                    qu[k] += b * gam / hss[k]
                else:
                    a = vx[k]
                    Hxx[k] -= np.outer(a, a) / hss[k]
                    qx[k] += a * gss[k] / hss[k]
            Pm = [None] * (N + 1); pv = [None] * (N + 1)
            K = [None] * N; kf = [None] * N
            Pm[N] = Hxx[N]; pv[N] = qx[N]
            if p.terminal_xy_equality:
                # augmentation of the terminal equality E dx_N = e: + rho/2 |E dx_N - e|^2 leaves the Newton step
                # unchanged (the term is constant on the constraint) but lets the stage-wise factorisation see the
                # curvature restricted to the constraint, where the exact Hessian only has to be positive definite
                Pm[N] = Pm[N].copy(); pv[N] = pv[N].copy()
                e_aug = prob.traj_ref[N, :2] - X[N, :2]
                Pm[N][0, 0] += opt.rho_eq; Pm[N][1, 1] += opt.rho_eq
                pv[N][:2] -= opt.rho_eq * e_aug
            pvv = [None] * (N + 1); kfv = [None] * N
            pvv[N] = np.zeros((nx, 2)); pvv[N][0, 0] = 1.0; pvv[N][1, 1] = 1.0    # E^T, E = [I2 0] (terminal xy equality)
            for k in range(N - 1, -1, -1):
                A, B = AB[k]
                PA = Pm[k + 1] @ A
                PB = Pm[k + 1] @ B
                pc = pv[k + 1] + Pm[k + 1] @ c[k]
                F = Hxx[k] + A.T @ PA
                G = Hux[k] + B.T @ PA
                Hh = Huu[k] + B.T @ PB
                gx_ = qx[k] + A.T @ pc
                gu_ = qu[k] + B.T @ pc
                L = _chol_checked(Hh, np.diag(Huu[k]) if use_exact else None, opt.piv_frac)
                if L is None:
                    return None
                K[k] = -_cho_solve(L, G)
                kf[k] = -_cho_solve(L, gu_)
                kfv[k] = -_cho_solve(L, B.T @ pvv[k + 1])
                pvv[k] = A.T @ pvv[k + 1] + G.T @ kfv[k]
                Pm[k] = F + G.T @ K[k]
                Pm[k] = 0.5 * (Pm[k] + Pm[k].T)
                pv[k] = gx_ + G.T @ kf[k]
            return Pm, pv, K, kf, hss, gss, vx, vxN, pvv, kfv
        # (not with the terminal equality: its full correction E dx_N = e is forced whatever the damping, the multipliers
        #  nu then grow like prox)
        def newton():
            """factorisation of the Newton matrix: exact Hessian, inertia-corrected where a pivot fails (or the round-3 ladder)"""
            nonlocal delta_last, prox_cur, delta_used
            delta_used = 0.0
            prox = prox_cur if (opt.prox and not p.terminal_xy_equality) else 0.0
            skip0 = opt.inertia and not p.terminal_xy_equality and opt.ic_skip0 > 0 and delta_prev_it > opt.ic_skip0
            fac = factor(opt.exact_hessian, prox) if (opt.exact_hessian and not skip0) else None
            if fac is None and opt.exact_hessian and opt.inertia and not p.terminal_xy_equality:
                delta = opt.delta0 if delta_last == 0.0 else max(opt.delta_min, delta_last * opt.kappa_dn)
                while True:
                    fac = factor(True, prox + delta)
                    if fac is not None or delta > 1e40:
                        break
                    delta = delta * (opt.kappa_first if delta_last == 0.0 else opt.kappa_up)
                if fac is not None:
                    delta_last = delta; delta_used = delta
                return fac
            if fac is None and opt.exact_hessian and opt.mid_fallback and not p.terminal_xy_equality and \
                    (prob.hs is None or len(prob.hs) == 0):
                fac = factor(True, prox, dyn_curv=False)
            if fac is None:
                fac = factor(False, prox)
                while fac is None and opt.prox and not p.terminal_xy_equality and prox_cur < opt.prox_max:
                    prox_cur = min(opt.prox_max, max(opt.prox0, prox_cur * opt.prox_up))
                    fac = factor(False, prox_cur)
            return fac

        def direction(fac):
            """search direction from a factorisation: roll-out, multiplier and row steps, fraction to the boundary"""
            Pm, pv, K, kf, hss, gss, vx, vxN, pvv, kfv = fac
            nu_new = np.zeros(2)
            if p.terminal_xy_equality:
                # interface_wholebody_qref.py:166-167: X[N,:2] == X_ref[N,:2].  The two multipliers enter the terminal
                # gradient linearly, so the direction is affine in them: roll out the nu = 0 solution and the two
                # sensitivities with the SAME feedback gains, then solve the 2x2 system E dx_N = e.
                d0 = np.zeros(nx); Dv = np.zeros((nx, 2))
                for k in range(N):
                    A, B = AB[k]
                    u0 = K[k] @ d0 + kf[k]; Uv = K[k] @ Dv + kfv[k]
                    d0 = A @ d0 + B @ u0 + c[k]; Dv = A @ Dv + B @ Uv
                e = prob.traj_ref[N, :2] - X[N, :2]
                nu_new = np.linalg.solve(Dv[:2, :], e - d0[:2])
                kf = [kf[k] + kfv[k] @ nu_new for k in range(N)]
            dX = np.zeros_like(X); dU = np.zeros_like(U)
            dX[0] = prob.x_init - X[0]
            for k in range(N):
                A, B = AB[k]
                dU[k] = K[k] @ dX[k] + kf[k]
                dX[k + 1] = A @ dX[k] + B @ dU[k] + c[k]
            lam_new = np.zeros_like(lam)
            for k in range(1, N + 1):
                lam_new[k] = -(Pm[k] @ dX[k] + pv[k] + pvv[k] @ nu_new)
            ds = np.zeros(N + 1)
            for k in range(N + 1):
                vdx = vx[k] @ dX[k]
                if k == N - 1:
                    vdx += vxN @ dX[N]
                ds[k] = -(gss[k] - vdx) / hss[k]
            dt = [None] * (N + 1); dz = [None] * (N + 1)
            for k in range(N + 1):
                dtk = np.zeros(len(rows[k])); dzk = np.zeros(len(rows[k]))
                for i, (h, jx, ju, ks, Hr) in enumerate(ev[k]):
                    jd = 0.0
                    if jx is not None: jd += jx @ dX[k]
                    if ju is not None: jd += ju @ dU[k]
                    if ks is not None: jd -= ds[ks]
                    dtk[i] = -rh[k][i] - jd
                    w = z[k][i] / t[k][i]
                    dzk[i] = mu / t[k][i] - z[k][i] - w * dtk[i]
                dt[k] = dtk; dz[k] = dzk
            tau = max(opt.tau_min, 1 - mu)
            ap = 1.0; ad = 1.0
            for k in range(N + 1):
                neg = dt[k] < 0
                if neg.any():
                    ap = min(ap, (-tau * t[k][neg] / dt[k][neg]).min())
                neg = dz[k] < 0
                if neg.any():
                    ad = min(ad, (-tau * z[k][neg] / dz[k][neg]).min())
            return dX, dU, ds, lam_new, nu_new, dt, dz, ap, ad

        fac = newton()
        if fac is None:
            status = 2
            break
        delta_prev_it = delta_used
        dX, dU, ds, lam_new, nu_new, dt, dz, ap, ad = direction(fac)
        if getattr(opt, "dense_check", False):
            _dense_check(prob, rows, ev, X, U, s, t, z, lam, mu, c, AB, gX, gU, gs, rh, dX, dU, ds, lam_new, opt,
                         Q2, P2, RW2)
        # ---------------- line search -------------------------------------
        phi0, th0 = _merit_parts(prob, rows, X, U, s, t, mu)
        dphi = (gX * dX).sum() + (gU * dU).sum() + gs @ ds
        for k in range(N + 1):
            dphi -= mu * (dt[k] / t[k]).sum()
        alpha = ap
        accepted = False
        z_done = False
        if opt.filter:
            if filt is None:
                filt = []
                th_max = 1e4 * max(1.0, th0)
                th_min = 1e-4 * max(1.0, th0)
            def acceptable(th, phi, a0):
                """filter and sufficient-decrease test of a trial point; a0: the step length of the switching / Armijo conditions"""
                ok_f = th < th_max and all(not (th >= fj[0] and phi >= fj[1]) for fj in filt)
                ftype = (dphi < 0 and th0 <= th_min and a0 * (-dphi) ** 2.3 > th0 ** 1.1)
                if ok_f:
                    if ftype:
                        return phi <= phi0 + 1e-8 * a0 * dphi + 1e-14 * abs(phi0)
                    if th <= (1 - 1e-5) * th0 or phi <= phi0 - 1e-5 * th0:
                        _filter_add(filt, ((1 - 1e-5) * th0, phi0 - 1e-5 * th0))
                        return True
                return False
            for lspass in range(2):
                alpha = ap
                for ls in range(opt.max_ls):
                    Xn = X + alpha * dX; Un = U + alpha * dU; sn = s + alpha * ds
                    tn = [t[k] + alpha * dt[k] for k in range(N + 1)]
                    phi, th = _merit_parts(prob, rows, Xn, Un, sn, tn, mu)
                    if acceptable(th, phi, alpha):
                        accepted = True
                        break
                    if ls == 0 and lspass == 0 and opt.soc_max > 0 and th >= th0 and th0 <= th_min and not p.terminal_xy_equality:
                        # second-order correction (Waechter & Biegler 2006, section 2.4; see oracle/mmpc_oracle.c): the row
                        # multipliers take their step first, then the direction is recomputed from x_k with
                        # c_soc = alpha c(x_k) + c(x_k + alpha d) in place of the constraint residuals
                        for k in range(N + 1):
                            z[k] = z[k] + ad * dz[k]
                            z[k] = np.minimum(np.maximum(z[k], mu / (opt.kappa_sigma * tn[k])), opt.kappa_sigma * mu / tn[k])
                        z_done = True
                        keep = (dX, dU, ds, lam_new, nu_new, dt, c, rh)
                        accC = c.copy(); accR = [r.copy() for r in rh]
                        a_prev, th_prev, alpha0 = alpha, th, alpha
                        for _ in range(opt.soc_max):
                            evn = _eval_all(prob, rows, Xn, Un, sn, 0)
                            accC = a_prev * accC + _defects(prob, Xn, Un)
                            accR = [a_prev * accR[k] + np.array([e[0] for e in evn[k]]) + tn[k] for k in range(N + 1)]
                            c = accC; rh = accR
                            fac2 = newton()
                            if fac2 is None:
                                break
                            dX, dU, ds, lam_new, nu_new, dt, dz2, ap2, ad2 = direction(fac2)
                            Xn = X + ap2 * dX; Un = U + ap2 * dU; sn = s + ap2 * ds
                            tn = [t[k] + ap2 * dt[k] for k in range(N + 1)]
                            phi2, th2 = _merit_parts(prob, rows, Xn, Un, sn, tn, mu)
                            if acceptable(th2, phi2, alpha0):
                                accepted = True
                                alpha = ap2
                                break
                            if th2 > 0.99 * th_prev:
                                break
                            a_prev, th_prev = ap2, th2
                        c, rh = keep[6], keep[7]
                        if accepted:
                            break
                        dX, dU, ds, lam_new, nu_new, dt = keep[:6]      # the line search goes on along the uncorrected direction
                    if ls < opt.max_ls - 1:
                        alpha *= 0.5
                if accepted or not filt:
                    break
                filt.clear()      # filter reset heuristic: the filter blocked every trial step
            dm = dphi
        else:
            if th0 > 1e-14:
                need = dphi / (0.9 * th0)
                if need > nu_merit:
                    nu_merit = need + 1.0
            if opt.nu_lam:
                nu_merit = max(nu_merit, opt.nu_lam * np.abs(lam_new).max())
            dm = dphi - nu_merit * th0
            for ls in range(opt.max_ls):
                Xn = X + alpha * dX; Un = U + alpha * dU; sn = s + alpha * ds
                tn = [t[k] + alpha * dt[k] for k in range(N + 1)]
                phi, th = _merit_parts(prob, rows, Xn, Un, sn, tn, mu)
                if phi + nu_merit * th <= phi0 + nu_merit * th0 + opt.eta * alpha * dm + 1e-13 * abs(phi0):
                    accepted = True
                    break
                alpha *= 0.5
        nfail += (not accepted)
        if opt.prox:
            a_px = ap if opt.prox_on_ap else alpha    # (the crawl the term is for is the one the fraction-to-boundary rule causes)
            nsmall = nsmall + 1 if a_px < opt.prox_lo else 0
            if a_px < opt.prox_lo and (nsmall >= opt.prox_need or prox_cur > 0.0):
                prox_cur = min(opt.prox_max, max(opt.prox0, prox_cur * opt.prox_up))
            elif alpha > opt.prox_hi:
                prox_cur = prox_cur / opt.prox_dn if prox_cur > opt.prox0 * 1e-3 else 0.0
        if verbose:
            print(f"      ap {ap:.3f} ad {ad:.3f} alpha {alpha:.4f} ls {ls} nu {nu_merit:.2e} dm {dm:.3e}")
        X = X + alpha * dX; U = U + alpha * dU; s = s + alpha * ds
        X[0] = prob.x_init
        for k in range(N + 1):
            t[k] = t[k] + alpha * dt[k]
            if not z_done:
                z[k] = z[k] + ad * dz[k]
        lam = lam + alpha * (lam_new - lam)
        nu_eq = nu_eq + alpha * (nu_new - nu_eq)
    return dict(X=X, U=U, s=s, status=status, iters=it, cost=nlp.cost(prob, X, U, s), E0=E0,
                lam=lam, mu=mu, nfail=nfail)


FCAP = 16


def _filter_add(filt, ent):
    """fixed-capacity filter: when full, the entry with the largest theta is replaced"""
    if len(filt) < FCAP:
        filt.append(ent)
    else:
        im = max(range(FCAP), key=lambda i: (filt[i][0], -i))
        filt[im] = ent


def _chol_reg(H, reg0):
    """Cholesky of a small SPD block; if a pivot is <= 0 add delta*I (delta = reg0, x10...)."""
    delta = 0.0
    n = 0
    while True:
        try:
            L = np.linalg.cholesky(H + delta * np.eye(H.shape[0]))
            return L, n
        except np.linalg.LinAlgError:
            delta = reg0 if delta == 0.0 else delta * 10
            n += 1


def _chol_checked(H, ref_diag, frac):
    """Cholesky; returns None if a pivot is not safely positive (exact-Hessian attempt)."""
    n = H.shape[0]
    L = np.zeros_like(H)
    for j in range(n):
        d = H[j, j] - L[j, :j] @ L[j, :j]
        lim = 0.0 if ref_diag is None else frac * ref_diag[j]
        if not (d > lim) or not np.isfinite(d):
            if ref_diag is None:
                raise FloatingPointError("GN Hessian not PD: %r" % d)
            return None
        L[j, j] = np.sqrt(d)
        for i in range(j + 1, n):
            L[i, j] = (H[i, j] - L[i, :j] @ L[j, :j]) / L[j, j]
    return L


def _cho_solve(L, B):
    y = np.linalg.solve(L, B)
    return np.linalg.solve(L.T, y)


def _dense_check(prob, rows, ev, X, U, s, t, z, lam, mu, c, AB, gX, gU, gs, rh, dX, dU, ds, lam_new, opt,
                 Q2, P2, RW2):
    """Debug: assemble the full (un-condensed) Newton KKT system densely and compare."""
    p = prob.par
    N, nx, nu = p.N, p.nx, p.nu
    nvar = (N + 1) * nx + N * nu + N + 1
    offU = (N + 1) * nx
    offS = offU + N * nu
    H = np.zeros((nvar, nvar)); g = np.zeros(nvar)
    for k in range(N + 1):
        xs = slice(k * nx, (k + 1) * nx)
        H[xs, xs] += Q2 if k < N else P2
        g[xs] += gX[k]
        H[offS + k, offS + k] += 2 * p.S
        g[offS + k] += gs[k]
        if k < N:
            us = slice(offU + k * nu, offU + (k + 1) * nu)
            H[us, us] += RW2
            g[us] += gU[k]
            if opt.exact_hessian:
                fxx, fux, fuu = nlp.f_hess_contract(p.kind, X[k], U[k], p.dt, lam[k + 1])
                H[xs, xs] -= fxx; H[us, xs] -= fux; H[xs, us] -= fux.T; H[us, us] -= fuu
        for i, (h, jx, ju, ks, Hr) in enumerate(ev[k]):
            J = np.zeros(nvar)
            if jx is not None: J[xs] = jx
            if ju is not None: J[offU + k * nu: offU + (k + 1) * nu] = ju
            if ks is not None: J[offS + ks] = -1
            w = z[k][i] / t[k][i]
            zh = mu / t[k][i] + w * rh[k][i]
            H += w * np.outer(J, J)
            g += J * zh
            if Hr is not None and opt.exact_hessian:
                H[xs, xs] += z[k][i] * Hr
    neq = (N + 1) * nx
    Jc = np.zeros((neq, nvar)); rc = np.zeros(neq)
    Jc[:nx, :nx] = np.eye(nx); rc[:nx] = X[0] - prob.x_init
    for k in range(N):
        A, B = AB[k]
        r = slice((k + 1) * nx, (k + 2) * nx)
        Jc[r, (k + 1) * nx:(k + 2) * nx] = np.eye(nx)
        Jc[r, k * nx:(k + 1) * nx] = -A
        Jc[r, offU + k * nu:offU + (k + 1) * nu] = -B
        rc[r] = -c[k]
    KKT = np.block([[H, Jc.T], [Jc, np.zeros((neq, neq))]])
    sol = np.linalg.solve(KKT, -np.concatenate([g, rc]))
    dw = sol[:nvar]; lamd = sol[nvar:].reshape(N + 1, nx)
    dXd, dUd, dsd = nlp.unpack(p, dw)
    print("   dense-check: dX %.2e dU %.2e ds %.2e lam %.2e  (|dU| %.2e, |lam| %.2e)" % (
        np.abs(dXd - dX).max(), np.abs(dUd - dU).max(), np.abs(dsd - ds).max(),
        np.abs(lamd[1:] - lam_new[1:]).max(), np.abs(dUd).max(), np.abs(lamd).max()))
