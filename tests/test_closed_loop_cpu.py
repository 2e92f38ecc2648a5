"""Host logic of the closed loop (task state machine, global plans, local references, controller classes) with the GPU
engine replaced by the CPU oracle (tests/oracle_engine.py).  The same scenario runs on the real engine in
tests/test_gpu_parity.py::test_closed_loop_demo_state_machine."""
import contextlib
import io
import itertools

import numpy as np

from tests import oracle_engine


import pytest


@pytest.mark.parametrize("as_written", [True, False], ids=["nlp_as_written", "intended_rows"])
def test_demo_scenario_2_state_machine_on_oracle(mm, monkeypatch, as_written):
    """as_written: the reference's own constructor call (two half-space planes -> quirk Q8 rows in the NLP); every tick's
    solution is re-checked against the extra rows by the controller.  intended_rows: faithful_convex=False."""
    oracle_engine.patch(monkeypatch, mm)
    dt, N = 0.1, 20
    obstacles = [mm.Obstacles(2.5, 3.0, 0.6), mm.Obstacles(2.5, 1.0, 0.6), mm.Obstacles(5 - 0.6, 5, 0.1)]   # demo_wholebody_qref.py:35-39
    r2 = 1 / np.sqrt(2)
    manip = [(np.array([2.5, 2, 0.35 + 0.606 + 0.333]), np.array([[r2, 0, r2]])),                          # :30-33
             (np.array([2.5, 2, 0.35 + 0.606 + 0.333]), np.array([[-r2, 0, r2]]))]
    target = np.array([5 - 0.6, 5, 0.606 + 0.333 + 0.5, -np.pi])                                          # :29
    ctrl = mm.MPCWholeBody(mm.MobileManipulator(dt), obstacles, manip, N=N, max_iter=2000) if as_written else \
        mm.MPCWholeBody(mm.MobileManipulator(dt), obstacles, manip, N=N, faithful_convex=False)
    world = mm.Interface(dt, 5, 2, np.zeros(9), target, ctrl, physical_sim=False)
    assert np.allclose(world.x_target, [5.0, 5.0, -np.pi, 0, 0, 0, 0, 0, 0])
    with contextlib.redirect_stdout(io.StringIO()):
        flag = world.run(max_steps=400)
    phases = [k for k, _ in itertools.groupby(world.flag_log)]
    assert flag == 'manipulate finish', (flag, world.mpc_step_counter, phases)
    assert phases == ['move', 'approach', 'rotate', 'manipulate']
    X = np.array(world.x_log)
    for o in obstacles:
        assert (np.hypot(X[:, 0] - o.x, X[:, 1] - o.y) >= o.radius + 0.4 - 5e-3).all()   # soft rows (S = 1e5): mm-level slack
    assert np.linalg.norm(world.current_joints_pose[:3] - target[:3]) <= 0.01
    assert np.abs(X[-1, :2] - world.x_target[:2]).max() < 0.02
    assert abs(ctrl.angleDiff(X[-1, 2], -np.pi)) < 0.5 * np.pi / 180 + 1e-3
    # weights were switched by the state machine (interface_wholebody_qref.py:211-215)
    assert np.allclose(np.diag(ctrl.Q_value), [500, 500, 500, 0, 0, 1, 1, 1, 1])
    if as_written:
        assert ctrl.q8_margin is not None and ctrl.q8_margin <= ctrl.Q8_TOL


def test_interface_rejects_simulator_branch(mm, monkeypatch):
    oracle_engine.patch(monkeypatch, mm)
    ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=10)
    import pytest
    with pytest.raises(NotImplementedError):
        mm.Interface(0.1, 5, 2, np.zeros(9), np.array([1, 0, 1.4, 0.0]), ctrl, physical_sim=True)
