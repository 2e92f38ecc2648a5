"""Arm inverse kinematics (robot_models/manipulator_3DoF.py:79-133): oracle pinned by the reference's known answer,
host build of the kernel's per-lane function against the oracle, GPU kernel through the C ABI."""
import numpy as np
import pytest

from oracle import ik, nlp
from tests import emu_helper

# the reference's own known answer: utils/numerical_solve.py:5 (target), :21-36 (x0 = 0, same objective and box),
# value quoted at robot_models/manipulator_3DoF.py:219 (6 significant digits, IPOPT tol 1e-8)
KAT_TARGET = np.array([0.7, 0.5])
KAT_Q = np.array([0.695168, -0.467009, 2.66495])


def _reachable_cases(n, seed=3):
    rng = np.random.default_rng(seed)
    qt = rng.uniform(ik.LO, ik.HI, (n, 3))
    t = np.array([[s.sum() for s in ik._segments(q)] for q in qt])
    q0 = rng.uniform(ik.LO, ik.HI, (n, 3))
    return q0, t


def test_oracle_ik_known_answer_of_the_reference():
    q, st, _ = ik.solve(np.zeros(3), KAT_TARGET)
    assert st == 0
    assert np.abs(q - KAT_Q).max() < 1e-5                      # the quoted digits
    # the target is out of reach (|target| = 0.8602 > 0.8579 = sum of the link lengths): unique minimiser, arm stretched
    e = nlp.arm_fk(q)[0]
    assert np.hypot(*KAT_TARGET) > np.hypot(e[0], e[2]) and abs(np.hypot(e[0] - 0.7, e[2] - 0.5) - 2.3395e-3) < 1e-6
    assert ik.kkt(q, KAT_TARGET) < 1e-12


def test_oracle_ik_objective_matches_forward_kinematics():
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.uniform(ik.LO, ik.HI); t = rng.uniform(-0.5, 0.8, 2)
        e = nlp.arm_fk(q)[0]
        assert abs(ik.objective(q, t) - ((e[0] - t[0]) ** 2 + (e[2] - t[1]) ** 2)) < 1e-14


def test_emu_ik_matches_oracle():
    q0, t = _reachable_cases(200)
    q0 = np.vstack([np.zeros((1, 3)), q0]); t = np.vstack([KAT_TARGET[None], t])
    r = emu_helper.ik_batch(q0, t)
    assert (r["status"] == 0).all()
    assert np.abs(r["q"][0] - KAT_Q).max() < 1e-5
    for b in range(q0.shape[0]):
        qo, st, it = ik.solve(q0[b], t[b])
        assert st == 0 and np.abs(r["q"][b] - qo).max() < 1e-9, (b, r["q"][b], qo)
        assert ((r["q"][b] >= ik.LO) & (r["q"][b] <= ik.HI)).all()
        assert ik.kkt(r["q"][b], t[b]) < 1e-8


def test_emu_ik_edge_cases():
    # start on the box corner, target at the arm base, target far away, start outside the box (clipped first)
    q0 = np.array([ik.LO, ik.HI, [0.0, 0.0, 0.0], [5.0, -5.0, 9.0]])
    t = np.array([[0.3, 0.3], [0.0, 0.0], [10.0, -3.0], [0.2, 0.5]])
    r = emu_helper.ik_batch(q0, t)
    for b in range(4):
        assert r["status"][b] == 0
        assert ((r["q"][b] >= ik.LO) & (r["q"][b] <= ik.HI)).all()
        assert ik.kkt(r["q"][b], t[b]) < 1e-8
        qo, _, _ = ik.solve(q0[b], t[b])
        assert np.abs(r["q"][b] - qo).max() < 1e-8


@pytest.mark.gpu
def test_gpu_ik_parity_and_reference_api(mm):
    q0, t = _reachable_cases(4096, seed=11)
    q0[0] = 0.0; t[0] = KAT_TARGET
    r = mm.ManipulatorPanda3DoF.inverse_transformation_batch(q0, t)
    assert (r["status"] == 0).all()
    assert np.abs(r["q"][0] - KAT_Q).max() < 1e-5
    # bit-for-bit the same algorithm as the host build of mmpc_ik.h up to libm differences: compare with the oracle
    for b in range(0, 4096, 37):
        qo, _, _ = ik.solve(q0[b], t[b])
        assert np.abs(r["q"][b] - qo).max() < 1e-8
    # size-independent properties over the whole batch: inside the box, endpoint on target, stationary
    assert ((r["q"] >= ik.LO) & (r["q"] <= ik.HI)).all()
    res = np.array([np.sqrt(ik.objective(r["q"][b], t[b])) for b in range(4096)])
    kk = np.array([ik.kkt(r["q"][b], t[b]) for b in range(4096)])
    assert kk.max() < 1e-8
    assert (res < 1e-7).mean() > 0.9        # some random starts end in a local minimum on the box boundary (as any local solver)
    # single-instance API of the reference: inverse_transformation(q_initial_guess, x_target=(x, 0, z))
    arm = mm.ManipulatorPanda3DoF(0.1)
    q = arm.inverse_transformation(np.zeros(3), np.array([0.7, 0.0, 0.5]))
    assert np.abs(q - KAT_Q).max() < 1e-5
    with pytest.raises(ValueError):
        arm.inverse_transformation(np.zeros(3), np.array([0.7, 0.0, 0.5, 0.0]))
