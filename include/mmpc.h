/*
 * mmpc.h - C ABI of the MI355X batched MPC solve engine (libmmpc.so).
 *
 * The reference (HsinyuG/mobile-manipulator-mpc) is pure Python: its hot path has no FFI.  The
 * boundary this library replaces is the Python controller object the closed-loop driver calls
 * (interface_wholebody_qref.py:134  `self.controller.solve(current_state, local_traj_ref,
 * local_u_ref)`), i.e. the public surface of controllers/mpc_wholebody_qref.py:6-331 and
 * controllers/mpc_base.py:6-226.  Each entry point below names the reference interface it
 * stands in for.  Plain pointers and sizes only; nothing throws across the ABI; every function
 * returns 0 on success or a negative MMPC_E_* code (text via mmpc_last_error()).
 *
 * Array conventions are the reference's: float64, C order, one record per problem instance:
 *   x_init[B][nx]  traj_ref[B][N+1][nx]  u_ref[B][N][nu]  obs[B][M][3] = (x, y, radius)
 *   (or obs[B][N+1][M][3] when cfg.obs_per_stage: per-stage centres for moving obstacles)
 *   X[B][N+1][nx]  U[B][N][nu]  s[B][N+1]
 * whole-body kind: nx=9 [x,y,psi,dx,dy,dpsi,q1,q2,q3], nu=5 [dV,dw,dq1,dq2,dq3]
 * (robot_models/mobile_manipulator.py:22,62-65); base kind: nx=6, nu=2 (robot_models/base.py:26).
 */
#ifndef MMPC_H
#define MMPC_H
#ifdef __cplusplus
extern "C" {
#endif

#define MMPC_KIND_WHOLEBODY 0 /* controllers/mpc_wholebody_qref.py:MPCWholeBody */
#define MMPC_KIND_BASE 1      /* controllers/mpc_base.py:MPCBase */
#define MMPC_KIND_WHOLEBODY_POSE 2 /* controllers/mpc_wholebody.py:MPCWholeBody - whole-body model, the state cost tracks the
                                      endpoint pose forward_tranformation(x)[0] = (x, y, z, psi): traj_ref[B][N+1][4], Q and P
                                      are 4x4 (:11-12,79-80,104-106); circle rows only; X is warm-started too (:134-139) */

#define MMPC_OK 0
#define MMPC_E_ARG (-1)         /* bad argument / unsupported size */
#define MMPC_E_HIP (-2)         /* HIP runtime error */
#define MMPC_E_NODEVICE (-3)    /* no gfx950 device visible */
#define MMPC_E_UNSUPPORTED (-4) /* feature not implemented */

/* per-instance solver status written to status[B] */
#define MMPC_STATUS_CONVERGED 0 /* scaled KKT error <= tol */
#define MMPC_STATUS_MAXITER 1
#define MMPC_STATUS_NUMERIC 2   /* NaN / non-PD even with the Gauss-Newton Hessian; also: NaN or inf in the instance's data (x_init,
                                   references, obstacles, u_last) - detected at the first evaluation, the other instances of the
                                   batch are not affected (opti.solve() raises for such an instance, mpc_wholebody_qref.py:315) */
#define MMPC_STATUS_SUSPENDED 3 /* iteration budget of the launch used up (mmpc_set_iteration_budget): X, U, s hold the current
                                   iterate, mmpc_resume_batch_device continues the solve */

/* Build-time structure of the NLP = constructor arguments of MPCWholeBody.__init__
 * (mpc_wholebody_qref.py:7-23: N, ulim, xlim, dulim, robot.dt) / MPCBase.__init__
 * (mpc_base.py:7-17).  +-INFINITY marks an absent bound (the reference uses ca.inf). */
typedef struct mmpc_config {
    int kind;          /* MMPC_KIND_* */
    int N;             /* horizon, 1 <= N <= 63 */
    int M;             /* circle obstacles per instance, 0 <= M <= 16 (len(obstacle_list)) */
    int obs_per_stage; /* 0: obs[B][M][3]; 1: obs[B][N+1][M][3] */
    int max_batch;     /* capacity of the device-side buffers: warm start, outputs of the host-pointer call, launch order and - for
                          horizons N >= 21 on a specialised kernel - one 8 (N nu (nx + 1) + N nu (nu - 1) / 2 + 64) byte block of
                          feedback gains per instance (14.9 KB at N = 30: 122 MB for 8192 instances) */
    int device;        /* HIP device ordinal */
    int max_iter;      /* interior-point iteration cap (reference passes ipopt.max_iter 2000, :280) */
    double dt;         /* robot.dt (demo_wholebody_qref.py:10) */
    double tol;        /* scaled KKT tolerance (ipopt tol / acceptable_tol 1e-8, :283) */
    double mu_init;    /* initial barrier parameter */
    double ulim[2][5]; /* [lo|hi][nu] */
    double xlim[2][9]; /* [lo|hi][nx]; psi entry +-INFINITY (mpc_base.py:16 stores 5 columns) */
    double dulim[2][5];
                       /* a limit of magnitude >= 1e19 (in particular +-INFINITY) is no row: IPOPT's nlp_lower/upper_bound_inf,
                          which the reference leaves at their defaults */
    int L;             /* half-space ("manipulation") obstacles, 0..8: len(obstacle_manipulation_list)
                          (mpc_wholebody_qref.py:10,39; demo_wholebody_qref.py:21-33); whole-body kind only */
    double halfspace[8][6]; /* per obstacle: point (3), outward normal (3).  One row per (stage, arm sample point):
                          -max_j n_j.((p_j - 0.03 n_j) - P_i) <= s_k, the INTENDED form of obsAvoidConvex (:57-89) */
    int as_written;    /* L >= 2: the NLP AS WRITTEN - obsAvoidConvex calls subject_to inside its plane loop and keeps one `constr`
                          matrix for all stages (:77-89,156), so besides the intended row it emits, per (stage k >= 1, point i,
                          j < L-1):  -max(c_{k,i,0..j}, c_{k-1,i,j+1..L-1}) <= s_k  (the entries j' > j still hold the PREVIOUS
                          stage's expressions; at k = 0 they hold free variables that can always satisfy the row).  These rows
                          tie x_k to x_{k-1}; the generic kernel carries them.  0: intended rows only */
} mmpc_config;

typedef struct mmpc_handle_s *mmpc_handle;

/* MPCWholeBody.__init__ / MPCBase.__init__ + reset() (mpc_wholebody_qref.py:25-46,142-285):
 * allocates device state, uploads the structure.  Weights start at the reference defaults
 * (:12-16 / mpc_base.py:11-14). */
int mmpc_create(const mmpc_config *cfg, mmpc_handle *out);
int mmpc_destroy(mmpc_handle h);

/* setWeight(Q,R,P,S,W) (mpc_wholebody_qref.py:119-139; mpc_base.py:96-112 where S is `M`):
 * row-major nx*nx / nu*nu matrices (Q, P: 4*4 for MMPC_KIND_WHOLEBODY_POSE); a NULL pointer keeps the current value;
 * S<0 keeps S. */
int mmpc_set_weights(mmpc_handle h, const double *Q, const double *R, const double *P, double S, const double *W);

/* The hard terminal equality the driver injects through controller.opti
 * (interface_wholebody_qref.py:166-167: X[N,:2] == X_ref[N,:2]). */
int mmpc_set_terminal_xy_equality(mmpc_handle h, int on);

/* reset(): clears the warm start (u_latest / x_guess = None, mpc_wholebody_qref.py:164-165). */
int mmpc_reset(mmpc_handle h);

/* solve(x_init, traj_ref, u_ref) for B instances (mpc_wholebody_qref.py:287-331, mpc_base.py:191-226).
 * Host pointers.  Uses the handle's warm start (u_latest, and x_guess for the base and pose-reference kinds) for
 * instances 0..B-1 and replaces it for the instances that CONVERGED - the reference assigns u_latest / x_guess after a
 * successful solve only (:329-330 follow the exception of :315), a failed instance keeps what it had.
 * x_init of the whole-body kind is clipped to xlim (:290-291) inside.
 * Any out_* may be NULL except out_u0.  out_u0[B][nu] = U*[0] (the reference's return value).
 * B = 0 (an empty batch, e.g. the shard of a rank beyond the end of the batch) is a no-op that returns MMPC_OK, here and in
 * the device-pointer calls; B > max_batch is MMPC_E_ARG. */
int mmpc_solve_batch(mmpc_handle h, int B, const double *x_init, const double *traj_ref, const double *u_ref,
                     const double *obs, double *out_u0, double *out_X, double *out_U, double *out_s, int *out_status,
                     int *out_iters, double *out_cost);

/* Same solve on DEVICE pointers (inputs already resident in HBM), asynchronous on `stream`
 * (a hipStream_t, passed as void*; NULL = default stream).  Stateless: u_last is the explicit
 * U_last parameter / U initial guess (:303,:310), x_guess the X initial guess for the base kind
 * (mpc_base.py:200; NULL = tile(x_init), :302).  status/iters are int32 arrays, cost/err float64. */
int mmpc_solve_batch_device(mmpc_handle h, int B, const double *d_x_init, const double *d_traj_ref,
                            const double *d_u_ref, const double *d_u_last, const double *d_x_guess,
                            const double *d_obs, double *d_X, double *d_U, double *d_s, int *d_status, int *d_iters,
                            double *d_cost, double *d_err, void *stream);

/* List launch (specialised kernels; no reference counterpart): the same solve for the instances d_list[0 .. *d_count - 1] only -
 * row indices into the B-row arrays, every index at most once; list and count live in DEVICE memory (the count is read by the
 * kernel: a caller that builds the list on the device never has to synchronise), `capacity` (<= B) bounds the count and is the
 * size of the grid.  An entry outside [0, B) is skipped (nothing can validate a device-side list on the host); a repeated entry is
 * the caller's error (two workgroups would solve the same rows).  Rows that are not listed are neither read nor written.  Instances start in list order (put the ones that
 * are expected to take longest first); the handle's schedule hint is not used and not updated.  With an iteration budget set the
 * listed instances that need more are suspended as in mmpc_solve_batch_device and mmpc_resume_batch_device (same B) continues
 * exactly them.  What it is for: a receding-horizon loop over many robots in which the robots whose solve is finished advance
 * to their next tick while the few suspended ones are still being continued on another stream (bench.py --config c5). */
int mmpc_solve_list_device(mmpc_handle h, int B, const int *d_list, const int *d_count, int capacity, const double *d_x_init,
                           const double *d_traj_ref, const double *d_u_ref, const double *d_u_last, const double *d_x_guess,
                           const double *d_obs, double *d_X, double *d_U, double *d_s, int *d_status, int *d_iters,
                           double *d_cost, double *d_err, void *stream);

/* warm start access (self.u_latest, mpc_wholebody_qref.py:165,330): host <-> device copies */
int mmpc_get_u_latest(mmpc_handle h, int B, double *u_latest);
int mmpc_set_u_latest(mmpc_handle h, int B, const double *u_latest);

/* ManipulatorPanda3DoF.inverse_transformation(q_initial_guess, x_target) for B instances
 * (robot_models/manipulator_3DoF.py:79-133; called by Interface.globalPlanManipulator, interface_wholebody_qref.py:286):
 * minimise (x_e(q)-x*)^2 + (z_e(q)-z*)^2 over the joint box of :123, started at q0.  Stateless, no handle:
 * q0[B][3], target_xz[B][2] = (x*, z*) in the arm base frame (the reference asserts y* == 0, :100), out_q[B][3];
 * out_status[B] (0: projected gradient <= 1e-8) and out_iters[B] may be NULL.  Errors: mmpc_last_error(NULL). */
int mmpc_ik_batch(int device, int B, const double *q0, const double *target_xz, double *out_q, int *out_status,
                  int *out_iters);
int mmpc_ik_batch_device(int device, int B, const double *d_q0, const double *d_target_xz, double *d_q, int *d_status,
                         int *d_iters, void *stream);

/* Opt-in warm start for the device-pointer entry point (no reference counterpart: the reference starts every solve from
 * U = U_last, X = tile(x_init), mu = IPOPT's default, mpc_wholebody_qref.py:301-303).  d_u_guess [B][N][nu] (device memory
 * that stays valid until it is replaced or cleared with NULL) becomes the initial U of the following
 * mmpc_solve_batch_device calls - separate from d_u_last, which stays the U_last PARAMETER of the cost and of the rate
 * bounds -, their d_x_guess the initial X, mu_init the initial barrier parameter.  A receding-horizon caller passes the
 * previous optimum shifted by one stage, the roll-out of the dynamics under it, and mu_init = 0.1: the same NLP, solved
 * in fewer than half the iterations.  mmpc_set_warm_start(h, NULL, 1.0) restores the default. */
int mmpc_set_warm_start(mmpc_handle h, const double *d_u_guess, double mu_init);

/* Launch order of the workgroups (= instances) of a batch.  A batch takes as long as its last wave, so the instances
 * that need the most interior-point iterations should start first.  mode 1 (default): longest-first by the iteration counts
 * of the handle's previous launch of the same B when there is one (a receding-horizon loop solves the same robots every
 * tick), else by an a-priori difficulty key computed from this batch's own data (how deep the reference path cuts into an
 * inflated obstacle disc); mode 2: always the a-priori key (no memory of earlier launches); mode 0: batch order.
 * Results never depend on the order.  No reference counterpart (the reference solves one instance at a time). */
int mmpc_set_schedule_hint(mmpc_handle h, int mode);

/* Iteration budget and continuation (specialised kernels only; no reference counterpart - IPOPT runs one instance to the
 * end).  One wave solves one instance, so a launch lasts as long as its slowest instance: a handful of instances that need
 * several hundred iterations (heavy tail of the interior-point iteration count) would hold the results of thousands of
 * finished ones.  With budget > 0, mmpc_solve_batch_device gives every instance at most `budget` iterations per launch;
 * an instance that is not converged by then writes its primal-dual state to the handle (18.7 KB at N=20, M=5) and ends
 * with status MMPC_STATUS_SUSPENDED, X/U/s holding its current iterate.  mmpc_resume_batch_device - same arguments as the
 * launch it continues, any stream - runs the suspended instances to the end (no budget) with one workgroup each;
 * the other instances' outputs are not touched.  A resumed solve executes exactly the iterations the uninterrupted one
 * would have: results are bitwise identical.  mmpc_suspended_count waits for the handle's launches and returns how many
 * instances the last budgeted launch left suspended.  budget = 0 (default) switches the feature off.
 * A continuation belongs to the budgeted launch directly before it on the handle (same B): after any other launch, a first
 * continuation, mmpc_reset or a change of the budget the save areas are stale and mmpc_resume_batch_device returns
 * MMPC_E_UNSUPPORTED instead of overwriting results with the continuation of an older problem. */
int mmpc_set_iteration_budget(mmpc_handle h, int budget);
int mmpc_resume_batch_device(mmpc_handle h, int B, const double *d_x_init, const double *d_traj_ref, const double *d_u_ref,
                             const double *d_u_last, const double *d_x_guess, const double *d_obs, double *d_X, double *d_U,
                             double *d_s, int *d_status, int *d_iters, double *d_cost, double *d_err, void *stream);
int mmpc_suspended_count(mmpc_handle h, int *count);

/* Streams and threads: a handle owns device state that its launches read and write (parameter block, schedule hint,
 * warm start).  Calls on one handle must come from one host thread at a time.  Launches may use different streams:
 * a launch on another stream than the handle's previous one is ordered after it (event wait), and the entry points
 * that change the parameter block or the warm start (mmpc_set_weights, mmpc_set_terminal_xy_equality,
 * mmpc_set_warm_start, mmpc_reset, mmpc_get/set_u_latest) wait for the handle's launches in flight first. */

/* bytes of LDS one problem instance occupies (one 64-lane workgroup) */
int mmpc_lds_bytes(mmpc_handle h);
/* problem instances (workgroups) the runtime keeps resident per compute unit for the kernel this handle launches
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor: registers allow 4 - one wave per SIMD -, LDS may allow fewer) */
int mmpc_problems_per_cu(mmpc_handle h);
const char *mmpc_last_error(mmpc_handle h);
const char *mmpc_version(void);

#ifdef __cplusplus
}
#endif
#endif
