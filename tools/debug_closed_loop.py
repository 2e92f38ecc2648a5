"""Development aid: demo scenario 2 in closed loop on the GPU (NLP as written); on a failed solve the inputs of that tick are
written to gpurun_out/fail_tick.npz for a replay on the host build of the kernel."""
import sys, os, contextlib, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mmpc_loader
mm = mmpc_loader.load()
dt, N = 0.1, 20
r2 = 1 / np.sqrt(2)
obstacles = [mm.Obstacles(2.5, 3.0, 0.6), mm.Obstacles(2.5, 1.0, 0.6), mm.Obstacles(5 - 0.6, 5, 0.1)]
manip = [(np.array([2.5, 2, 0.35 + 0.606 + 0.333]), np.array([[r2, 0, r2]])), (np.array([2.5, 2, 0.35 + 0.606 + 0.333]), np.array([[-r2, 0, r2]]))]
target = np.array([5 - 0.6, 5, 0.606 + 0.333 + 0.5, -np.pi])
ctrl = mm.MPCWholeBody(mm.MobileManipulator(dt), obstacles, manip, N=N)
eng = ctrl._engine
orig = eng.solve_batch
log = []
def wrapped(x, t, u, o):
    ul = eng.get_u_latest(1)
    r = orig(x, t, u, o)
    log.append((int(r["status"][0]), int(r["iters"][0])))
    if r["status"][0] != 0:
        np.savez(os.path.join(ROOT, "gpurun_out", "fail_tick.npz"), x=x, t=t, u=u, o=o, ul=ul, Q=np.asarray(ctrl.Q_value, float),
                 P=np.asarray(ctrl.P_value, float), R=np.asarray(ctrl.R_value, float), W=np.asarray(ctrl.W_value, float),
                 teq=np.array(int(teq_state[0])), tick=len(log))
    return r
eng.solve_batch = wrapped
teq_state = [False]
orig_teq = eng.set_terminal_xy_equality
def teq_wrapped(on):
    teq_state[0] = bool(on)
    return orig_teq(on)
eng.set_terminal_xy_equality = teq_wrapped
world = mm.Interface(dt, 5, 2, np.zeros(9), target, ctrl, physical_sim=False)
try:
    with contextlib.redirect_stdout(io.StringIO()):
        flag = world.run(max_steps=400)
except Exception as e:
    flag = "EXC " + str(e)
print(flag, "ticks", len(log), "phase log tail", world.flag_log[-3:], "iters max", max(i for _, i in log), "last", log[-3:])
