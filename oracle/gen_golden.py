"""Generates tests/golden/*.npz from the reference (runs ONLY in the build container, where
/root/reference is mounted; the fixtures - data, not code - are committed and travel).

Two sources, both the reference's own code executed on numbers:

 1. utils/dh_to_kinematics.py (pure sympy, runs unmodified): its DH table is evaluated
    numerically -> positions of joint 2 / joint 3 / endpoint (frames 3 / 5 / 7) and the
    endpoint Jacobian, for random joint angles.  Pins the arm FK restated in oracle/nlp.py.

 2. robot_models/{base,manipulator_3DoF,mobile_manipulator}.py and the un-bound methods
    MPCWholeBody.angleDiff / obsAvoid (controllers/mpc_wholebody_qref.py:49-54,92-117),
    MPCBase.angleDiff / obsAvoid, and Interface.globalPlan2D / calcLocalRefTraj /
    calcLocalRefPose (interface_wholebody_qref.py:247-266,353-410).  Those files
    `import casadi` only for element-wise sin/cos/sqrt/fmod/horzcat/vertcat/if_else on what
    are plain floats here; CasADi 3.6.4 (requirements.txt:9) is not in this image, so this
    script puts a ~20-line numeric stand-in module named `casadi` (numpy element-wise
    functions, defined below) on sys.path for the duration of the run.  Nothing that reaches
    ca.Opti / ca.nlpsol (reset(), solve(), inverse_transformation) can be executed this way:
    the solver output of the reference stays UNPINNED (see oracle/nlp.py header).

Usage:  python oracle/gen_golden.py        (writes tests/golden/reference_model.npz, fk_dh_sympy.npz)
"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

SHIM = '''
import numpy as np
pi = np.pi
inf = np.inf
sin, cos, sqrt, fmod = np.sin, np.cos, np.sqrt, np.fmod
class MX:  # never instantiated on the numeric path
    pass
def horzcat(*a):
    a = [np.atleast_1d(np.asarray(x, dtype=float)) for x in a]
    return np.hstack(a) if all(x.ndim <= 1 for x in a) else np.hstack([np.atleast_2d(x) for x in a])
def vertcat(*a):
    return np.vstack([np.atleast_2d(np.asarray(x, dtype=float)) for x in a])
def norm_2(x):
    return np.linalg.norm(x)
def if_else(c, a, b):
    return a if c else b
def mtimes(*a):
    a = a[0] if len(a) == 1 and isinstance(a[0], (list, tuple)) else a
    out = a[0]
    for m in a[1:]:
        out = out @ m
    return out
'''


def gen_dh(rng):
    import sympy as sp
    sys.path.insert(0, os.path.join(REF, "utils"))
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):       # the script prints its symbolic results on import
        dh = importlib.import_module("dh_to_kinematics")
    q1, q2, q3 = dh.q1, dh.q2, dh.q3
    consts = {dh.a2: 0.316, dh.a3: 0.0825, dh.a5: 0.384, dh.a6: 0.088, dh.a7: 0.107}   # manipulator_3DoF.py:18-22
    Ts = dh.Ts
    f = [sp.lambdify((q1, q2, q3), Ts[i][:3, 3].subs(consts), "numpy") for i in (3, 5, 7)]
    Jf = sp.lambdify((q1, q2, q3), dh.jacobian_matrix.subs(consts), "numpy")
    Q = rng.uniform(-np.pi, np.pi, (64, 3))
    pos = np.array([[np.asarray(fi(*q), float).reshape(3) for fi in f] for q in Q])     # (64, 3 frames, 3)
    J = np.array([np.asarray(Jf(*q), float) for q in Q])                                # (64, 6, 3)
    np.savez(os.path.join(OUT, "fk_dh_sympy.npz"), q=Q, joint2=pos[:, 0], joint3=pos[:, 1], endpoint=pos[:, 2], jac=J)
    print("fk_dh_sympy.npz:", Q.shape, pos.shape, J.shape)


def gen_models(rng):
    tmp = tempfile.mkdtemp(prefix="casadi_numeric_standin_")
    with open(os.path.join(tmp, "casadi.py"), "w") as fh:
        fh.write(SHIM)
    sys.path[:0] = [tmp, REF]
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    from robot_models.mobile_manipulator import MobileManipulator
    from robot_models.base import Base
    from robot_models.obstacles import Obstacles
    from controllers.mpc_wholebody_qref import MPCWholeBody
    from controllers.mpc_base import MPCBase
    mm = MobileManipulator(0.1)
    K = 64
    X = rng.uniform(-2, 2, (K, 9)); X[:, 6:] = rng.uniform(-np.pi, np.pi, (K, 3))
    U = rng.uniform(-2, 2, (K, 5))
    f9 = np.array([np.asarray(mm.f_kinematics(x.copy(), u.copy()), float).reshape(-1) for x, u in zip(X, U)])
    f6 = np.array([np.asarray(Base(0.1).f_kinematics(x[:6].copy(), u[:2].copy()), float).reshape(-1) for x, u in zip(X, U)])
    fk = [mm.forward_tranformation(x.copy()) for x in X]
    pe = np.array([np.asarray(a[0], float).reshape(-1) for a in fk]); j2 = np.array([np.asarray(a[1], float).reshape(-1) for a in fk])
    j3 = np.array([np.asarray(a[2], float).reshape(-1) for a in fk])
    afk = [mm.manipulator.forward_tranformation(x[6:].copy()) for x in X]
    ae = np.array([np.asarray(a[0], float).reshape(-1) for a in afk])
    # obsAvoid rows (un-bound method; needs only base_radius)
    obs = [Obstacles(2.5, 3.0, 0.6), Obstacles(2.5, 1.0, 0.6), Obstacles(5 - 0.6, 5, 0.1)]        # demo_wholebody_qref.py:40-44
    ns = types.SimpleNamespace(base_radius=0.4)
    g_wb = np.array([[float(v) for v in MPCWholeBody.obsAvoid(ns, obs, x)] for x in X])
    nsb = types.SimpleNamespace(base_radius=lambda: 0.4)
    g_b = np.array([[float(v) for v in MPCBase.obsAvoid(nsb, obs, x[:6])] for x in X])
    A = rng.uniform(-10, 10, (200, 2))
    A[:6] = [[-3.14, 3.14], [3.0, -3.0], [0.5, 0.2], [np.pi, -np.pi], [0.0, 0.0], [-0.1, 0.1]]
    ad_wb = np.array([float(MPCWholeBody.angleDiff(None, a, b)) for a, b in A])
    ad_b = np.array([float(MPCBase.angleDiff(None, a, b)) for a, b in A])
    # self-collision rows exactly as reset() writes them (mpc_wholebody_qref.py:213-222), on numbers
    selfrows = []
    for x in X:
        pose_e, x2, x3 = mm.forward_tranformation(x.copy())
        e = np.asarray(pose_e, float).reshape(-1)[:3]; x2 = np.asarray(x2, float).reshape(-1); x3 = np.asarray(x3, float).reshape(-1)
        chk = [np.zeros(3), x2 / 2, x2, (x2 + x3) / 2]
        selfrows.append([0.05 - np.linalg.norm(c - e) for c in chk])
    out = dict(X=X, U=U, f_wholebody=f9, f_base=f6, pose_endpoint=pe, pos_joint2=j2, pos_joint3=j3, arm_endpoint=ae,
               obs=np.array([[o.x, o.y, o.radius] for o in obs]), obsavoid_wholebody=g_wb, obsavoid_base=g_b,
               angle_pairs=A, anglediff_wholebody=ad_wb, anglediff_base=ad_b, selfcol_rows=np.array(selfrows))
    # Interface planning helpers (caller side of the boundary), with the simulator import stubbed out
    sys.modules["simulation"] = types.ModuleType("simulation")
    sys.modules["simulation.albert_robot"] = types.ModuleType("simulation.albert_robot")
    from interface_wholebody_qref import Interface
    ctrl = types.SimpleNamespace(N=20, robot_model=mm, angleDiff=lambda a, b: MPCWholeBody.angleDiff(None, a, b))
    it = Interface(0.1, 5, 2, np.zeros(9), np.array([4.4, 5, 1.439, -np.pi]), ctrl, physical_sim=False)
    it.globalPlan2D()
    out["plan_traj_ref"] = np.asarray(it.traj_ref, float); out["plan_u_ref"] = np.asarray(it.u_ref, float)
    out["plan_x_target"] = np.asarray(it.x_target, float)
    # calcLocalRefTraj(distance_index): nearest global-reference point to current_state[distance_index], window of
    # N+1 rows, tail padded with the last row (interface_wholebody_qref.py:353-396)
    states = np.zeros((5, 9)); states[:, :2] = [[0.0, 0.0], [0.3, 0.2], [2.2, 2.6], [4.9, 5.0], [5.0, 5.0]]
    states[:, 2] = [0.0, 0.4, 1.0, 3.0, -3.0]
    wins, uwins, pose_wins = [], [], []
    for st in states:
        it.current_state = st.copy()
        it.calcLocalRefTraj([0, 1])
        wins.append(np.asarray(it.local_traj_ref, float).copy()); uwins.append(np.asarray(it.local_u_ref, float).copy())
        it.calcLocalRefPose()                       # :398-410 (continuous-psi reference)
        pose_wins.append(np.asarray(it.local_traj_ref, float).copy())
    out["window_states"] = states; out["window_traj"] = np.array(wins); out["window_u"] = np.array(uwins)
    out["pose_window_traj"] = np.array(pose_wins)
    np.savez(os.path.join(OUT, "reference_model.npz"), **out)
    print("reference_model.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20240114)
    gen_dh(rng)
    gen_models(rng)
