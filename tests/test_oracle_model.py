"""CPU: the numpy restatement of the reference's kinematics / constraint rows (oracle/nlp.py):
analytic derivatives vs complex-step differentiation, closed forms vs the literal expressions,
and the C port (oracle/mmpc_oracle.c) vs numpy."""
import numpy as np
import pytest

from oracle import nlp, coracle


def _cs_grad(f, x, n):
    g = np.zeros((n, len(x)))
    for i in range(len(x)):
        xx = x.astype(complex)
        xx[i] += 1e-30j
        g[:, i] = np.imag(np.atleast_1d(f(xx))) / 1e-30
    return g


def test_self_collision_rows_closed_form_and_derivatives():
    rng = np.random.default_rng(1)
    for _ in range(50):
        x = rng.uniform(-2, 2, 9)
        x[6:] = rng.uniform(-3, 3, 3)
        for i in range(4):
            h, g, H = nlp.selfcol_row(x, i)
            assert abs(h - nlp.selfcol_row_direct(x, i)) < 1e-14      # literal mpc_wholebody_qref.py:213-222
            g2 = _cs_grad(lambda z: nlp.selfcol_row(z, i, order=0), x, 1)[0]
            assert np.abs(g - g2).max() < 1e-13
            Hn = _cs_grad(lambda z: nlp.selfcol_row(z, i, order=1)[1], x, 9)
            assert np.abs(H - Hn).max() < 1e-12


def test_circle_row_derivatives():
    rng = np.random.default_rng(2)
    for _ in range(50):
        x = rng.uniform(-2, 2, 9)
        o = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(0.1, 0.6)])
        g0, g, H = nlp.circle_row(x, o, 9)
        assert abs(g0 - ((o[2] + 0.4) - np.hypot(x[0] - o[0], x[1] - o[1]))) < 1e-15   # :53, base.py:15
        assert np.abs(g - _cs_grad(lambda z: nlp.circle_row(z, o, 9, order=0), x, 1)[0]).max() < 1e-13
        assert np.abs(H - _cs_grad(lambda z: nlp.circle_row(z, o, 9, order=1)[1], x, 9)).max() < 1e-12


def test_dynamics_jacobians_and_curvature():
    rng = np.random.default_rng(3)
    for kind, nx, nu in (("wholebody", 9, 5), ("base", 6, 2)):
        for _ in range(20):
            x, u = rng.uniform(-2, 2, nx), rng.uniform(-2, 2, nu)
            A, B = nlp.f_jac(kind, x, u, 0.1)
            assert np.abs(A - _cs_grad(lambda z: nlp.f_dyn(kind, z, u, 0.1), x, nx)).max() < 1e-14
            assert np.abs(B - _cs_grad(lambda z: nlp.f_dyn(kind, x, z, 0.1), u, nx)).max() < 1e-14
            lam = rng.uniform(-1, 1, nx)
            Hxx, Hux, Huu = nlp.f_hess_contract(kind, x, u, 0.1, lam)

            def gradL(z):
                A_, B_ = nlp.f_jac(kind, z[:nx], z[nx:], 0.1)
                return np.concatenate([A_.T @ lam, B_.T @ lam])
            z0 = np.concatenate([x, u])
            Hn = np.zeros((nx + nu, nx + nu))
            for j in range(nx + nu):
                e = np.zeros(nx + nu); e[j] = 1e-6
                Hn[:, j] = (gradL(z0 + e) - gradL(z0 - e)) / 2e-6
            assert np.abs(np.block([[Hxx, Hux.T], [Hux, Huu]]) - Hn).max() < 1e-8


def test_fk_sample_from_survey():
    """sample captured from the reference's robot_models run under a numeric casadi stand-in."""
    x = np.array([1, 2, .3, .1, -.2, .05, .2, -1, 1.5])
    u = np.array([.5, -.3, .1, -.2, .3])
    f = nlp.wholebody_f(x, u, 0.1)
    assert np.allclose(f, [1.01, 1.98, .305, .14876682, -.18472399, .02, .21, -1.02, 1.53], atol=5e-9)
    e, j2, j3 = nlp.wholebody_fk(x)
    assert np.allclose(e, [1.55441403, 2.17150036, 1.37213419, .3], atol=5e-9)
    assert np.allclose(j2, [1.1305324, 2.0403784, 1.23231082], atol=5e-9)
    assert np.allclose(j3, [1.44389089, 2.13731154, 1.44834942], atol=5e-9)


def test_ik_known_answer():
    """utils/numerical_solve.py:5,35 / manipulator_3DoF.py:219: q=(0.695168,-0.467009,2.66495) <-> (x,z)=(0.7,0.5)."""
    e, _, _ = nlp.arm_fk(np.array([0.695168, -0.467009, 2.66495]))
    assert abs(e[0] - 0.7) < 3e-3 and abs(e[2] - 0.5) < 3e-3 and e[1] == 0.0


def test_angle_diff_known_values():
    """SURVEY §8(c) values from the reference's angleDiff (mpc_wholebody_qref.py:92-117)."""
    assert abs(nlp.angle_diff(-3.14, 3.14) - 0.0031853) < 1e-7
    assert abs(nlp.angle_diff(3.0, -3.0) - (-0.2831853)) < 1e-7
    assert abs(nlp.angle_diff(0.5, 0.2) - 0.3) < 1e-15


def test_c_port_matches_numpy_model():
    rng = np.random.default_rng(4)
    for _ in range(50):
        x, u = rng.uniform(-2, 2, 9), rng.uniform(-2, 2, 5)
        assert np.abs(coracle.f("wholebody", 0.1, x, u) - nlp.wholebody_f(x, u, 0.1)).max() < 1e-15
        assert np.abs(coracle.f("base", 0.1, x[:6], u[:2]) - nlp.base_f(x[:6], u[:2], 0.1)).max() < 1e-15
        e, j2, j3 = coracle.fk(x)
        e_, j2_, j3_ = nlp.wholebody_fk(x)
        assert max(np.abs(e - e_).max(), np.abs(j2 - j2_).max(), np.abs(j3 - j3_).max()) < 1e-15
