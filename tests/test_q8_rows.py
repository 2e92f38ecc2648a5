"""CPU: the half-space rows of obsAvoidConvex AS WRITTEN (mpc_wholebody_qref.py:77-89,156): the host-side
evaluation the controller uses (mmpc_amd/controllers/_q8.py) against the oracle's restatement, the Jacobian of the extra rows
against finite differences, and a hand-made trajectory that satisfies the intended rows but not the as-written ones."""
import numpy as np

import mmpc_loader
from oracle import nlp

R2 = 1 / np.sqrt(2)
HS2 = np.array([[2.5, 2, 0.35 + 0.606 + 0.333, R2, 0, R2], [2.5, 2, 0.35 + 0.606 + 0.333, -R2, 0, R2]])    # demo scenario 2 (:30-33)
HS3 = np.array([[2.5, 2, 1.3, R2, 0, R2], [2.5, 2, 1.3, -R2, 0, R2], [2.5, 2, 1.5, 0, 0, 1.0]])           # three planes (mmax branch, :87)


def _traj(rng, N=6):
    X = np.zeros((N + 1, 9))
    X[:, 0] = np.linspace(1.8, 3.2, N + 1) + rng.normal(0, 0.02, N + 1)
    X[:, 1] = 2.0 + rng.normal(0, 0.05, N + 1)
    X[:, 2] = rng.normal(0, 0.2, N + 1)
    X[:, 6] = rng.uniform(-0.5, 0.5, N + 1); X[:, 7] = rng.uniform(-2, -0.3, N + 1); X[:, 8] = rng.uniform(0.3, 2.5, N + 1)
    return X


def test_host_check_equals_oracle_rows():
    q8 = mmpc_loader.load().controllers._q8
    rng = np.random.default_rng(4)
    for hs in (HS2, HS3):
        X = _traj(rng); s = rng.uniform(0, 0.05, X.shape[0])
        c = q8.plane_values(X, hs)
        for k in range(X.shape[0]):
            for i in range(6):
                assert np.allclose(c[k, i], nlp.halfspace_planes(X[k], i, hs), atol=1e-14)
                assert abs(-c[k, i].max() - nlp.halfspace_row_direct(X[k], i, hs)) < 1e-14        # intended row = last as-written row
        rows = q8.as_written_extra_rows(X, s, hs)
        par = nlp.WholeBodyParams(N=X.shape[0] - 1)
        prob = nlp.Problem(par, X[0], X, np.zeros((par.N, 5)), np.zeros((par.N, 5)), np.zeros((0, 3)), hs, as_written=True)
        assert rows.shape == (par.N, 6, len(hs) - 1)
        for k in range(1, par.N + 1):
            for i in range(6):
                for j in range(len(hs) - 1):
                    assert abs(rows[k - 1, i, j] - nlp.q8_extra_row(prob, X, s, k, i, j)) < 1e-14
    assert q8.as_written_extra_rows(X, s, HS2[:1]).shape[-1] == 0                                  # L = 1: nothing extra


def test_extra_rows_enter_the_flat_nlp_with_a_correct_jacobian():
    rng = np.random.default_rng(5)
    X = _traj(rng); N = X.shape[0] - 1
    U = rng.normal(0, 0.1, (N, 5)); s = rng.uniform(0, 0.05, N + 1)
    par = nlp.WholeBodyParams(N=N)
    base = nlp.Problem(par, X[0], X, np.zeros((N, 5)), np.zeros((N, 5)), np.zeros((0, 3)), HS3)
    full = nlp.Problem(par, X[0], X, np.zeros((N, 5)), np.zeros((N, 5)), np.zeros((0, 3)), HS3, as_written=True)
    h0, h1 = nlp.ineq_rows(base, X, U, s), nlp.ineq_rows(full, X, U, s)
    assert len(h1) == len(h0) + N * 6 * 2 and np.array_equal(h1[:len(h0)], h0)
    J = nlp.ineq_jac(full, X, U, s)
    w = nlp.pack(X, U, s)
    for col in rng.choice(len(w), 40, replace=False):
        e = np.zeros(len(w)); e[col] = 1e-6
        fd = (nlp.ineq_rows(full, *nlp.unpack(par, w + e)) - nlp.ineq_rows(full, *nlp.unpack(par, w - e))) / 2e-6
        assert np.abs(fd - J[:, col]).max() < 1e-6


def test_ridge_crossing_violates_only_the_as_written_rows():
    """An arm point that is on the plane-0 side at stage k-1 and on the plane-1 side at stage k satisfies the intended row at
    both stages, but row (k, i, 0) as written reads max(c_{k,i,0}, c_{k-1,i,1}) < 0."""
    q8 = mmpc_loader.load().controllers._q8
    X = np.zeros((2, 9)); X[:, 1] = 2.0; X[:, 6:] = [0.3, -1.2, 1.6]          # endpoint 0.64 m ahead of the base, 6 cm below the ridge
    X[0, 0], X[1, 0] = 1.72, 2.0                                  # the endpoint crosses the ridge at x = 2.5 in one step
    c = q8.plane_values(X, HS2)
    i = 5                                                          # endpoint
    assert c[0, i, 0] > 0 > c[0, i, 1] and c[1, i, 1] > 0 > c[1, i, 0]
    s = np.zeros(2)
    assert (-c[:, i].max(axis=-1) - s).max() <= 0                  # the intended rows of this point hold at both stages
    assert q8.as_written_extra_rows(X, s, HS2)[0, i, 0] > 0        # the as-written row of stage 1 does not
