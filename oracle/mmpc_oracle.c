/*
 * mmpc_oracle.c - scalar C (fp64) restatement of the MPC hot path, CPU only.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file, and only as the
 * checker / reported CPU baseline ("port").  The product library
 * (mobile-manipulator-mpc_amd/csrc) never links or calls it.
 *
 * What it restates (paths relative to /root/reference):
 *   - the NLP the reference builds in MPCWholeBody.reset()
 *     (controllers/mpc_wholebody_qref.py:142-285) and MPCBase.reset()
 *     (controllers/mpc_base.py:114-189): dynamics robot_models/base.py:17-26,
 *     robot_models/manipulator_3DoF.py:189-191, robot_models/mobile_manipulator.py:57-75;
 *     forward kinematics manipulator_3DoF.py:10-77 + mobile_manipulator.py:17-55;
 *     circle rows mpc_wholebody_qref.py:49-54; self-collision rows :219-222,:261-265
 *     (incl. the s[N-1] quirk of :265); boxes :203-205,:245;
 *   - the per-tick protocol of solve() (mpc_wholebody_qref.py:287-331, mpc_base.py:191-226).
 * The reference hands that NLP to CasADi 3.6.4 -> IPOPT -> MUMPS (requirements.txt:9), which
 * is not in this image; the solver below is the build's own interior-point method - the same
 * algorithm as oracle/ipm_numpy.py and the HIP kernels, written independently of the latter as
 * straightforward dense scalar code.  Since round 4 it follows IPOPT (Waechter & Biegler 2006) in its termination test
 * (eq. 5-6: complementarity over s_c), its inertia correction (Algorithm IC) and its second-order correction (section 2.4);
 * constants and the three deviations are listed in DESIGN.md section 3.
 *
 * PARITY STATUS: model functions pinned by tests/golden (reference robot_models + sympy DH);
 * solve() output of the reference is UNPINNED (no executable IPOPT here, no reference tests);
 * certified by KKT residuals (oracle/nlp.py:kkt_certificate) and scipy SLSQP (oracle/xcheck.py).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NXM 9
#define NUM 5
#define NSM 65          /* max stages (N+1) */
#define MMX 16          /* max circle obstacles */
#define LMX 8           /* max half-space obstacles */
#define NQ8M (6 * (LMX - 1))   /* extra half-space rows per stage of the NLP as written (quirk Q8) */
#define RMX (2 * NUM + 2 * NXM + MMX + 4 + 6 + NQ8M)
#define FCAP 16
#define PROX0 100.0
#define PROX_LO 0.05
#define PROX_MAX 1e4
#define KAPPA_SIGMA 1e10
#define RHO_EQ 1e4     /* augmentation weight of the terminal-xy equality inside the factorisation */

/* experiment knobs (tools / tests only; the defaults are the algorithm the kernels implement) */
/* [0] complementarity scaled by s_c (1) or s_d (0) | [1] inertia correction (1) or the round-3 Hessian ladder (0) | [2] second-order
 * corrections per iteration | [3] 1: no proximal term | [6..9] inertia correction: first delta, first growth, growth, decay |
 * [18] proximal trigger on the fraction-to-boundary step (1) or on the accepted step (0) | [20] second-order correction only where
 * theta(x_k) <= [20] theta_min (0: always) | [25] an iteration whose predecessor needed a correction above this starts from
 * the corrected matrix at once (0: always try the uncorrected matrix first, as IPOPT does) */
static double LAB[32] = {1, 1, 2, 0, 0, 0, 1, 4, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0.1};
static long long LAB_NTRIAL = 0, LAB_NFACT = 0, LAB_NSOC = 0, LAB_NSOCOK = 0, LAB_NSOCFAIL = 0;
void mmpc_oracle_set_lab(int i, double v) { if (i >= 0 && i < 32) LAB[i] = v; }
double mmpc_oracle_get_lab(int i) { return i == 104 ? (double)LAB_NTRIAL : i == 100 ? (double)LAB_NFACT : i == 101 ? (double)LAB_NSOC : i == 102 ? (double)LAB_NSOCOK : i == 103 ? (double)LAB_NSOCFAIL : (i >= 0 && i < 32 ? LAB[i] : 0.0); }

typedef struct {
    int kind;            /* 0 whole-body (nx=9,nu=5), 1 base-only (nx=6,nu=2) */
    int N, M;
    int obs_per_stage;   /* 0: obs[M][3]; 1: obs[N+1][M][3] */
    int terminal_xy_eq;
    double dt;
    double Q[NXM * NXM], P[NXM * NXM], R[NUM * NUM], W[NUM * NUM], S;
    double ulim[2][NUM], xlim[2][NXM], dulim[2][NUM];
    double tol, mu_init;
    int max_iter;
    int L;               /* half-space ("manipulation") obstacles, demo_wholebody_qref.py:21-33 */
    double hs[LMX][6];   /* point (3), normal (3) */
    int pose_ref;        /* controllers/mpc_wholebody.py: the state cost tracks the endpoint pose (x,y,z,psi); Q, P are 4x4
                            (leading dimension 4), traj_ref is [N+1][4]; no self-collision / half-space rows */
    int as_written;      /* L >= 2: also the L-1 extra rows per (stage >= 1, arm point) that obsAvoidConvex emits as written
                            (quirk Q8, mpc_wholebody_qref.py:77-89,156): -max(c_{k,i,0..j}, c_{k-1,i,j+1..L-1}) <= s_k */
} oracle_cfg;

/* robot_models/manipulator_3DoF.py:18-22, mobile_manipulator.py:14-15, base.py:15,
 * mpc_wholebody_qref.py:43 */
static const double A2 = 0.316, A3 = 0.0825, A5 = 0.384, A6 = 0.088, A7 = 0.107;
static const double BX = -0.007, BZ = 0.606 + 0.333, BASE_R = 0.4, SELF_R = 0.05;

/* ------------------------------------------------------------------ model */
static void f_dyn(int kind, double dt, const double *x, const double *u, double *xn) {
    /* base.py:19-26 */
    double c = cos(x[2]), s = sin(x[2]);
    xn[0] = x[0] + dt * x[3];
    xn[1] = x[1] + dt * x[4];
    xn[2] = x[2] + dt * x[5];
    xn[3] = x[3] + dt * (u[0] * c - x[4] * x[5]);
    xn[4] = x[4] + dt * (u[0] * s + x[3] * x[5]);
    xn[5] = x[5] + dt * u[1];
    if (kind == 0) { /* manipulator_3DoF.py:190 */
        xn[6] = x[6] + dt * u[2];
        xn[7] = x[7] + dt * u[3];
        xn[8] = x[8] + dt * u[4];
    }
}

static void f_jac(int kind, double dt, const double *x, const double *u, double A[NXM][NXM], double B[NXM][NUM]) {
    int nx = kind == 0 ? 9 : 6;
    memset(A, 0, sizeof(double) * NXM * NXM);
    memset(B, 0, sizeof(double) * NXM * NUM);
    for (int i = 0; i < nx; i++) A[i][i] = 1.0;
    double c = cos(x[2]), s = sin(x[2]);
    A[0][3] = dt; A[1][4] = dt; A[2][5] = dt;
    A[3][2] = -dt * u[0] * s; A[3][4] = -dt * x[5]; A[3][5] = -dt * x[4];
    A[4][2] = dt * u[0] * c;  A[4][3] = dt * x[5];  A[4][5] = dt * x[3];
    B[3][0] = dt * c; B[4][0] = dt * s; B[5][1] = dt;
    if (kind == 0) { B[6][2] = dt; B[7][3] = dt; B[8][4] = dt; }
}

/* planar segments of the arm, manipulator_3DoF.py:29-73 collapsed */
static void arm_segments(const double *q, double drho[3], double dzet[3]) {
    double a = q[0] - q[1], b = q[0] - q[1] - q[2];
    double s1 = sin(q[0]), c1 = cos(q[0]), sA = sin(a), cA = cos(a), sB = sin(b), cB = cos(b);
    drho[0] = A2 * s1 + A3 * c1;   dzet[0] = A2 * c1 - A3 * s1;
    drho[1] = -A3 * cA + A5 * sA;  dzet[1] = A3 * sA + A5 * cA;
    drho[2] = A6 * cB - A7 * sB;   dzet[2] = -A6 * sB - A7 * cB;
}

/* world FK, mobile_manipulator.py:17-55: out e[4], j2[3], j3[3] */
void mmpc_oracle_fk(const double *x, double *e, double *j2, double *j3) {
    double dr[3], dz[3];
    arm_segments(x + 6, dr, dz);
    double c = cos(x[2]), s = sin(x[2]);
    double r2 = dr[0], z2 = dz[0], r3 = r2 + dr[1], z3 = z2 + dz[1], re = r3 + dr[2], ze = z3 + dz[2];
    j2[0] = x[0] + (r2 + BX) * c; j2[1] = x[1] + (r2 + BX) * s; j2[2] = z2 + BZ;
    j3[0] = x[0] + (r3 + BX) * c; j3[1] = x[1] + (r3 + BX) * s; j3[2] = z3 + BZ;
    e[0] = x[0] + (re + BX) * c;  e[1] = x[1] + (re + BX) * s;  e[2] = ze + BZ; e[3] = x[2];
}

void mmpc_oracle_f(int kind, double dt, const double *x, const double *u, double *xn) { f_dyn(kind, dt, x, u, xn); }

static const double SEGC[3][3] = {{1, 0, 0}, {1, -1, 0}, {1, -1, -1}};
static const double AB_SELF[4][2] = {{0, 0}, {0.5, 0}, {1, 0}, {0.5, 0.5}};
static const int YIDX[6] = {0, 1, 2, 6, 7, 8};

/* self-collision row i: h = 0.05 - ||alpha j2 + beta j3 - e||  (mpc_wholebody_qref.py:219-222);
 * g6 = dh/d(x,y,psi,q1,q2,q3) */
static double self_row(const double *x, int i, double *g6) {
    double al = AB_SELF[i][0], be = AB_SELF[i][1], kap = al + be - 1.0;
    double cm[3] = {kap, be - 1.0, -1.0};
    double dr[3], dz[3];
    arm_segments(x + 6, dr, dz);
    double R = kap * BX, Z = kap * BZ;
    for (int m = 0; m < 3; m++) { R += cm[m] * dr[m]; Z += cm[m] * dz[m]; }
    double px = x[0], py = x[1], c = cos(x[2]), s = sin(x[2]);
    double C = px * c + py * s, D = -px * s + py * c;
    double mm = kap * kap * (px * px + py * py) + 2 * kap * R * C + R * R + Z * Z;
    double n = sqrt(mm);
    if (g6) {
        double Ri[3] = {0, 0, 0}, Zi[3] = {0, 0, 0};
        for (int m = 0; m < 3; m++)
            for (int j = 0; j < 3; j++) { Ri[j] += cm[m] * dz[m] * SEGC[m][j]; Zi[j] -= cm[m] * dr[m] * SEGC[m][j]; }
        double gm[6];
        gm[0] = 2 * kap * kap * px + 2 * kap * R * c;
        gm[1] = 2 * kap * kap * py + 2 * kap * R * s;
        gm[2] = 2 * kap * R * D;
        for (int j = 0; j < 3; j++) gm[3 + j] = 2 * (kap * C + R) * Ri[j] + 2 * Z * Zi[j];
        for (int j = 0; j < 6; j++) g6[j] = -gm[j] / (2 * n);
    }
    return SELF_R - n;
}

/* half-space row for sample point i of [j2/2, j2, (j2+j3)/2, j3, (j3+e)/2, e] (mpc_wholebody_qref.py:57-89,216-217):
 *   h = -max_j n_j.((pi_j - 0.03 n_j) - P_i(x));   one row per (k,i) - the intended formulation, see oracle/nlp.py */
static const double HS_PTS[6][3] = {{0.5, 0, 0}, {1, 0, 0}, {0.5, 0.5, 0}, {0, 1, 0}, {0, 0.5, 0.5}, {0, 0, 1}};
/* the same for ONE plane range [j0, j1): -max_{j0 <= j < j1} c_{i,j}(x), derivatives of the plane that attains it */
static double hs_row_range(const oracle_cfg *cfg, const double *x, int i, int j0, int j1, double *g6, double *h10);
static double hs_row(const oracle_cfg *cfg, const double *x, int i, double *g6, double *h10) {
    return hs_row_range(cfg, x, i, 0, cfg->L, g6, h10);
}
static double hs_row_range(const oracle_cfg *cfg, const double *x, int i, int j0, int j1, double *g6, double *h10) {
    double al = HS_PTS[i][0], be = HS_PTS[i][1], ga = HS_PTS[i][2], sig = al + be + ga;
    double cm[3] = {sig, be + ga, ga}, dr[3], dz[3];
    arm_segments(x + 6, dr, dz);
    double R = sig * BX, Z = sig * BZ;
    for (int m = 0; m < 3; m++) { R += cm[m] * dr[m]; Z += cm[m] * dz[m]; }
    double c = cos(x[2]), s = sin(x[2]);
    double P[3] = {sig * x[0] + R * c, sig * x[1] + R * s, Z};
    double best = 0; int jb = -1;
    for (int j = j0; j < j1; j++) {
        const double *pi = cfg->hs[j], *n = cfg->hs[j] + 3;
        double v = 0; for (int a = 0; a < 3; a++) v += n[a] * ((pi[a] - 0.03 * n[a]) - P[a]);
        if (jb < 0 || v > best) { best = v; jb = j; }
    }
    if (g6) {
        const double *n = cfg->hs[jb] + 3;
        double Rm[3] = {0, 0, 0}, Zm[3] = {0, 0, 0};
        for (int m = 0; m < 3; m++) for (int j = 0; j < 3; j++) { Rm[j] += cm[m] * dz[m] * SEGC[m][j]; Zm[j] -= cm[m] * dr[m] * SEGC[m][j]; }
        g6[0] = sig * n[0]; g6[1] = sig * n[1]; g6[2] = n[0] * (-R * s) + n[1] * (R * c);
        for (int j = 0; j < 3; j++) g6[3 + j] = n[0] * Rm[j] * c + n[1] * Rm[j] * s + n[2] * Zm[j];
        if (h10) {
            /* second derivatives of n.P_i(x) w.r.t. (psi, q1, q2, q3), packed lower triangle; d2 seg/d theta^2 = -seg, so
             * R_qq = -sum_t cm_t dr_t D_t D_t^T (entries are +-Zm), Z_qq likewise with dz (entries are -+Rm) */
            double nc = n[0] * c + n[1] * s, nt = -n[0] * s + n[1] * c;
            double Rqq[6] = {Zm[0], Zm[1], -Zm[1], Zm[2], -Zm[2], -Zm[2]};   /* (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) */
            double Zqq[6] = {-Rm[0], -Rm[1], Rm[1], -Rm[2], Rm[2], Rm[2]};
            h10[0] = -nc * R;
            h10[1] = nt * Rm[0]; h10[2] = nc * Rqq[0] + n[2] * Zqq[0];
            h10[3] = nt * Rm[1]; h10[4] = nc * Rqq[1] + n[2] * Zqq[1]; h10[5] = nc * Rqq[2] + n[2] * Zqq[2];
            h10[6] = nt * Rm[2]; h10[7] = nc * Rqq[3] + n[2] * Zqq[3]; h10[8] = nc * Rqq[4] + n[2] * Zqq[4]; h10[9] = nc * Rqq[5] + n[2] * Zqq[5];
        }
    }
    return -best;
}

/* circle row: g = (r+0.4) - dist  (mpc_wholebody_qref.py:53); grad (2), hess (xx,xy,yy) */
static double circ_row(const double *x, const double *o, double *g2, double *h3) {
    double dx = x[0] - o[0], dy = x[1] - o[1], d = sqrt(dx * dx + dy * dy);
    if (g2) { g2[0] = -dx / d; g2[1] = -dy / d; }
    if (h3) {
        double nx_ = dx / d, ny_ = dy / d;
        h3[0] = -(1 - nx_ * nx_) / d; h3[1] = nx_ * ny_ / d; h3[2] = -(1 - ny_ * ny_) / d;
    }
    return (o[2] + BASE_R) - d;
}

static double angle_diff(double a, double b) { /* mpc_base.py:56-94 */
    a = fmod(a + M_PI, 2 * M_PI) - M_PI;
    b = fmod(b + M_PI, 2 * M_PI) - M_PI;
    double d = a - b;
    if (a * b >= 0) return d;
    if (a > b) return d <= M_PI ? d : d - 2 * M_PI;
    return d > -M_PI ? d : d + 2 * M_PI;
}
double mmpc_oracle_angle_diff(double a, double b) { return angle_diff(a, b); }

/* ------------------------------------------------------------ solver state */
typedef struct {
    const oracle_cfg *cfg;
    int nx, nu, N, M, nrow; /* nrow = slots per stage */
    const double *xinit, *xref, *uref, *ulast, *obs;
    double X[NSM][NXM], U[NSM][NUM], s[NSM], lam[NSM][NXM];
    double t[NSM][RMX], z[NSM][RMX], bnd[NSM][RMX];
    int act[NSM][RMX];
    /* evaluation */
    double h[NSM][RMX];
    double gcirc[NSM][MMX][2], hcirc[NSM][MMX][3], gself[NSM][4][6], ghs[NSM][6][6], hhs[NSM][6][10];
    int nhs;
    /* rows of the NLP as written (quirk Q8): row e = i (L-1) + j of stage k >= 1 reads planes 0..j at x_k and planes j+1..L-1 at
     * x_{k-1}; q8br = 1 when the previous stage's entry attains the max (the row's gradient then acts on x_{k-1}) */
    int nq8, q8br[NSM][NQ8M];
    double gq8[NSM][NQ8M][6], hq8[NSM][NQ8M][10], vq[NSM][NXM];
    double A[NSM][NXM][NXM], B[NSM][NXM][NUM], c[NSM][NXM];
    double gX[NSM][NXM], gU[NSM][NUM], gs[NSM];
    /* QP */
    double Hxx[NSM][NXM][NXM], Hux[NSM][NUM][NXM], Huu[NSM][NUM][NUM], qx[NSM][NXM], qu[NSM][NUM];
    double hss[NSM], gss[NSM], vx[NSM][NXM], vxN[NXM];
    double K[NSM][NUM][NXM], kf[NSM][NUM];
    double pvv[NSM][NXM][3], kfv[NSM][NUM][3], nu_eq[2], nu_new[2];   /* terminal xy equality (interface_wholebody_qref.py:166-167): columns 0, 1 */
    /* column 2: the slack s_{N-1} kept as a BORDER variable of the stage-wise system when it reaches back to x_{N-2} (NLP as
     * written): it then touches x_{N-2}, x_{N-1} and (quirk Q1) x_N - three stages, no stage-wise elimination is exact */
    int sig; double dsig, sig_den, dxs[NSM][NXM];
    double dX[NSM][NXM], dU[NSM][NUM], ds[NSM], lamn[NSM][NXM];
    double dt_[NSM][RMX], dz[NSM][RMX];
    double Q2[NXM][NXM], P2[NXM][NXM], RW2[NUM][NUM];
    int dbg_k, dbg_r; double dbg_t, dbg_dt, dbg_z;
} work;

static const double *obs_at(const work *w, int k, int m) {
    return w->cfg->obs_per_stage ? w->obs + ((size_t)k * w->M + m) * 3 : w->obs + (size_t)m * 3;
}
/* slot layout: [0,nu) u-lo, [nu,2nu) u-hi, then x-lo (nx), x-hi (nx), circ (M), self (4) */
#define SL_ULO(w, j) (j)
#define SL_UHI(w, j) ((w)->nu + (j))
#define SL_XLO(w, j) (2 * (w)->nu + (j))
#define SL_XHI(w, j) (2 * (w)->nu + (w)->nx + (j))
#define SL_CIRC(w, m) (2 * (w)->nu + 2 * (w)->nx + (m))
#define SL_SELF(w, i) (2 * (w)->nu + 2 * (w)->nx + (w)->M + (i))
#define SL_HS(w, i) (2 * (w)->nu + 2 * (w)->nx + (w)->M + 4 + (i))
#define SL_Q8(w, e) (2 * (w)->nu + 2 * (w)->nx + (w)->M + 4 + (w)->nhs + (e))

static int slack_idx(const work *w, int k) { return k < w->N - 1 ? k : w->N - 1; } /* :265 quirk */

#define BOUND_PUSH 1e-2
static double bound_push(double v, double lo, double hi) {
    double lo2 = lo + BOUND_PUSH, hi2 = hi - BOUND_PUSH;
    if (lo2 > hi2) lo2 = hi2 = 0.5 * (lo + hi);
    return v < lo2 ? lo2 : (v > hi2 ? hi2 : v);
}

static void setup_rows(work *w) {
    const oracle_cfg *c = w->cfg;
    for (int k = 0; k <= w->N; k++) {
        for (int r = 0; r < w->nrow; r++) { w->act[k][r] = 0; w->bnd[k][r] = 0; }
        if (k < w->N)
            for (int j = 0; j < w->nu; j++) {
                /* merged box: ulim (:203) and u_last + dulim (:205) bound the same variable */
                double lo = c->ulim[0][j], hi = c->ulim[1][j];
                double l2 = w->ulast[k * w->nu + j] + c->dulim[0][j], h2 = w->ulast[k * w->nu + j] + c->dulim[1][j];
                if (l2 > lo) lo = l2;
                if (h2 < hi) hi = h2;
                if (isfinite(lo)) { w->act[k][SL_ULO(w, j)] = 1; w->bnd[k][SL_ULO(w, j)] = lo; }
                if (isfinite(hi)) { w->act[k][SL_UHI(w, j)] = 1; w->bnd[k][SL_UHI(w, j)] = hi; }
            }
        if (k >= 1)
            for (int j = 0; j < w->nx; j++) {
                if (isfinite(c->xlim[0][j])) { w->act[k][SL_XLO(w, j)] = 1; w->bnd[k][SL_XLO(w, j)] = c->xlim[0][j]; }
                if (isfinite(c->xlim[1][j])) { w->act[k][SL_XHI(w, j)] = 1; w->bnd[k][SL_XHI(w, j)] = c->xlim[1][j]; }
            }
        for (int m = 0; m < w->M; m++) w->act[k][SL_CIRC(w, m)] = 1;
        if (c->kind == 0 && !c->pose_ref) for (int i = 0; i < 4; i++) w->act[k][SL_SELF(w, i)] = 1;
        for (int i = 0; i < w->nhs; i++) w->act[k][SL_HS(w, i)] = 1;
        if (k >= 1) for (int e = 0; e < w->nq8; e++) w->act[k][SL_Q8(w, e)] = 1;   /* (at k = 0 the stale entries are free variables: no row) */
    }
}

/* row values at (X,U,s); with_deriv also fills gcirc/hcirc/gself */
static void eval_rows(work *w, double X[NSM][NXM], double U[NSM][NUM], const double *s, double h[NSM][RMX], int with_deriv) {
    for (int k = 0; k <= w->N; k++) {
        if (k < w->N)
            for (int j = 0; j < w->nu; j++) {
                if (w->act[k][SL_ULO(w, j)]) h[k][SL_ULO(w, j)] = w->bnd[k][SL_ULO(w, j)] - U[k][j];
                if (w->act[k][SL_UHI(w, j)]) h[k][SL_UHI(w, j)] = U[k][j] - w->bnd[k][SL_UHI(w, j)];
            }
        for (int j = 0; j < w->nx; j++) {
            if (w->act[k][SL_XLO(w, j)]) h[k][SL_XLO(w, j)] = w->bnd[k][SL_XLO(w, j)] - X[k][j];
            if (w->act[k][SL_XHI(w, j)]) h[k][SL_XHI(w, j)] = X[k][j] - w->bnd[k][SL_XHI(w, j)];
        }
        for (int m = 0; m < w->M; m++)
            h[k][SL_CIRC(w, m)] = circ_row(X[k], obs_at(w, k, m), with_deriv ? w->gcirc[k][m] : 0,
                                            with_deriv ? w->hcirc[k][m] : 0) - s[k];
        if (w->cfg->kind == 0 && !w->cfg->pose_ref)
            for (int i = 0; i < 4; i++)
                h[k][SL_SELF(w, i)] = self_row(X[k], i, with_deriv ? w->gself[k][i] : 0) - s[slack_idx(w, k)];
        for (int i = 0; i < w->nhs; i++) h[k][SL_HS(w, i)] = hs_row(w->cfg, X[k], i, with_deriv ? w->ghs[k][i] : 0, with_deriv ? w->hhs[k][i] : 0) - s[k];
        if (k >= 1 && w->nq8) {
            const int L = w->cfg->L;
            for (int i = 0; i < 6; i++) for (int j = 0; j < L - 1; j++) {
                const int e = i * (L - 1) + j;
                double gc[6], hc[10], gp[6], hp[10];
                const double vc = hs_row_range(w->cfg, X[k], i, 0, j + 1, with_deriv ? gc : 0, with_deriv ? hc : 0);         /* -max_{j' <= j} c_k */
                const double vp = hs_row_range(w->cfg, X[k - 1], i, j + 1, L, with_deriv ? gp : 0, with_deriv ? hp : 0);   /* -max_{j' > j} c_{k-1} */
                const int br = vp < vc;   /* -max(a, b) = min(-a, -b) */
                h[k][SL_Q8(w, e)] = (br ? vp : vc) - s[k];
                if (with_deriv) {
                    w->q8br[k][e] = br;
                    memcpy(w->gq8[k][e], br ? gp : gc, sizeof(gc));
                    memcpy(w->hq8[k][e], br ? hp : hc, sizeof(hc));
                }
            }
        }
    }
}

static void state_err(const work *w, const double *xk, const double *ref, double *e) {
    for (int j = 0; j < w->nx; j++) e[j] = xk[j] - ref[j];
    if (w->cfg->kind == 1) e[2] = angle_diff(xk[2], ref[2]); /* mpc_base.py:148 */
}

/* endpoint pose E = forward_tranformation(x)[0] = (x + R cos psi, y + R sin psi, Z, psi) (mobile_manipulator.py:36-53), its
 * Jacobian over y = (x, y, psi, q1, q2, q3) and the pieces of its second derivatives (see oracle/nlp.py: endpoint_pose) */
static void endpoint_pose(const double *x, double E[4], double J[4][6], double *Rout, double Rm[3], double Zm[3], double *cs) {
    double dr[3], dz[3];
    arm_segments(x + 6, dr, dz);
    double R = BX + dr[0] + dr[1] + dr[2], Z = BZ + dz[0] + dz[1] + dz[2], c = cos(x[2]), s = sin(x[2]);
    E[0] = x[0] + R * c; E[1] = x[1] + R * s; E[2] = Z; E[3] = x[2];
    if (!J) return;
    Rm[0] = dz[0] + dz[1] + dz[2]; Rm[1] = -(dz[1] + dz[2]); Rm[2] = -dz[2];
    Zm[0] = -(dr[0] + dr[1] + dr[2]); Zm[1] = dr[1] + dr[2]; Zm[2] = dr[2];
    memset(J, 0, sizeof(double) * 24);
    J[0][0] = 1; J[0][2] = -R * s; J[1][1] = 1; J[1][2] = R * c; J[3][2] = 1;
    for (int a = 0; a < 3; a++) { J[0][3 + a] = Rm[a] * c; J[1][3 + a] = Rm[a] * s; J[2][3 + a] = Zm[a]; }
    *Rout = R; cs[0] = c; cs[1] = s;
}

/* state term of stage k: e^T W e (W = Q or P).  grad (nx), Hgn = Gauss-Newton Hessian, Hcv = remaining curvature (zero unless
 * the cost goes through the forward kinematics, controllers/mpc_wholebody.py:79-80,104-106).  Any output may be NULL. */
static double state_cost(const work *w, int k, const double *xk, double *grad, double Hgn[NXM][NXM], double Hcv[NXM][NXM]) {
    const oracle_cfg *c = w->cfg;
    int nx = w->nx;
    const double *Wt = k < w->N ? c->Q : c->P;
    if (!c->pose_ref) {
        double e[NXM], q = 0;
        state_err(w, xk, w->xref + k * nx, e);
        for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) q += e[i] * Wt[i * nx + j] * e[j];
        if (grad) for (int i = 0; i < nx; i++) { double v = 0; for (int j = 0; j < nx; j++) v += (Wt[i * nx + j] + Wt[j * nx + i]) * e[j]; grad[i] = v; }
        if (Hgn) for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) { Hgn[i][j] = Wt[i * nx + j] + Wt[j * nx + i]; if (Hcv) Hcv[i][j] = 0; }
        return q;
    }
    double E[4], J[4][6], R = 0, Rm[3] = {0, 0, 0}, Zm[3] = {0, 0, 0}, cs[2] = {1, 0}, e[4], v[4], q = 0;
    endpoint_pose(xk, E, (grad || Hgn) ? J : 0, &R, Rm, Zm, cs);
    for (int i = 0; i < 4; i++) e[i] = E[i] - w->xref[k * 4 + i];
    for (int i = 0; i < 4; i++) { v[i] = 0; for (int j = 0; j < 4; j++) { q += e[i] * Wt[i * 4 + j] * e[j]; v[i] += (Wt[i * 4 + j] + Wt[j * 4 + i]) * e[j]; } }
    if (grad) { memset(grad, 0, sizeof(double) * nx); for (int a = 0; a < 6; a++) for (int i = 0; i < 4; i++) grad[YIDX[a]] += J[i][a] * v[i]; }
    if (Hgn) {
        for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) { Hgn[i][j] = 0; if (Hcv) Hcv[i][j] = 0; }
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
            double h = 0;
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) h += J[i][a] * (Wt[i * 4 + j] + Wt[j * 4 + i]) * J[j][b];
            Hgn[YIDX[a]][YIDX[b]] = h;
        }
        if (Hcv) {
            /* sum_c v_c d2E_c over (psi, q1, q2, q3): same closed form as the half-space rows with n = (v0, v1, v2) */
            double nc = v[0] * cs[0] + v[1] * cs[1], nt = -v[0] * cs[1] + v[1] * cs[0];
            double Rqq[3][3] = {{Zm[0], Zm[1], Zm[2]}, {Zm[1], -Zm[1], -Zm[2]}, {Zm[2], -Zm[2], -Zm[2]}};
            double Zqq[3][3] = {{-Rm[0], -Rm[1], -Rm[2]}, {-Rm[1], Rm[1], Rm[2]}, {-Rm[2], Rm[2], Rm[2]}};
            Hcv[2][2] = -nc * R;
            for (int a = 0; a < 3; a++) {
                Hcv[2][6 + a] = Hcv[6 + a][2] = nt * Rm[a];
                for (int b = 0; b < 3; b++) Hcv[6 + a][6 + b] = nc * Rqq[a][b] + v[2] * Zqq[a][b];
            }
        }
    }
    return q;
}

static double cost_fn(const work *w, double X[NSM][NXM], double U[NSM][NUM], const double *s) {
    /* mpc_wholebody_qref.py:199-201,227,242,270 */
    const oracle_cfg *c = w->cfg;
    int nx = w->nx, nu = w->nu;
    double J = 0, a[NUM], b[NUM];
    (void)nx;
    for (int k = 0; k <= w->N; k++) {
        J += state_cost(w, k, X[k], 0, 0, 0);
        J += c->S * s[k] * s[k];
        if (k < w->N) {
            for (int j = 0; j < nu; j++) { a[j] = U[k][j] - w->uref[k * nu + j]; b[j] = U[k][j] - w->ulast[k * nu + j]; }
            for (int i = 0; i < nu; i++) for (int j = 0; j < nu; j++)
                J += a[i] * c->R[i * nu + j] * a[j] + b[i] * c->W[i * nu + j] * b[j];
        }
    }
    return J;
}

/* barrier objective and l1 infeasibility at a trial point */
static void merit_parts(work *w, double X[NSM][NXM], double U[NSM][NUM], const double *s, double t[NSM][RMX],
                        double mu, double *phi, double *th) {
    static __thread double h[NSM][RMX];
    eval_rows(w, X, U, s, h, 0);
    double f = cost_fn(w, X, U, s), bar = 0, theta = 0, xn[NXM];
    for (int k = 0; k < w->N; k++) {
        f_dyn(w->cfg->kind, w->cfg->dt, X[k], U[k], xn);
        for (int j = 0; j < w->nx; j++) theta += fabs(xn[j] - X[k + 1][j]);
    }
    if (w->cfg->terminal_xy_eq) for (int j = 0; j < 2; j++) theta += fabs(X[w->N][j] - w->xref[w->N * w->nx + j]);
    for (int k = 0; k <= w->N; k++)
        for (int r = 0; r < w->nrow; r++)
            if (w->act[k][r]) { theta += fabs(h[k][r] + t[k][r]); bar -= mu * log(t[k][r]); }
    *phi = f + bar;
    *th = theta;
}

/* Cholesky of n x n (row-major, leading dim NUM); returns 0 if a pivot is not > 0 */
static int chol(double H[NUM][NUM], int n, double L[NUM][NUM]) {
    memset(L, 0, sizeof(double) * NUM * NUM);
    for (int j = 0; j < n; j++) {
        double d = H[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
        if (!(d > 0.0) || !isfinite(d)) return 0;
        L[j][j] = sqrt(d);
        for (int i = j + 1; i < n; i++) {
            double v = H[i][j];
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
            L[i][j] = v / L[j][j];
        }
    }
    return 1;
}
static void chol_solve(double L[NUM][NUM], int n, double *b) {
    for (int i = 0; i < n; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= L[i][k] * b[k]; b[i] = v / L[i][i]; }
    for (int i = n - 1; i >= 0; i--) { double v = b[i]; for (int k = i + 1; k < n; k++) v -= L[k][i] * b[k]; b[i] = v / L[i][i]; }
}

/* stage QP assembly + s Schur complement + Riccati factorisation.  returns 0 on a failed pivot.
 * use_exact 2: exact Lagrangian Hessian; 1: the same without the curvature of the dynamics, lam^T d2f (the middle rung of
 * the fallback ladder: keeps the curvature of the constraint rows and of the cost, which plain Gauss-Newton steps overshoot
 * without); 0: Gauss-Newton */
static int factor(work *w, double mu, int use_exact, double prox) {
    const oracle_cfg *c = w->cfg;
    __atomic_fetch_add(&LAB_NFACT, 1, __ATOMIC_RELAXED);
    int nx = w->nx, nu = w->nu, N = w->N;
    for (int k = 0; k <= N; k++) { w->hss[k] = 2 * c->S; w->gss[k] = w->gs[k]; memset(w->vx[k], 0, sizeof(w->vx[k])); memset(w->vq[k], 0, sizeof(w->vq[k])); }
    memset(w->vxN, 0, sizeof(w->vxN));
    /* (two sweeps over the stages: a row of stage k whose previous-stage entry attains the max adds to stage k-1's blocks) */
    for (int k = 0; k <= N; k++) {
        {
            double Hcv[NXM][NXM];
            state_cost(w, k, w->X[k], 0, w->Hxx[k], Hcv);
            for (int i = 0; i < nx; i++) { for (int j = 0; j < nx; j++) if (use_exact) w->Hxx[k][i][j] += Hcv[i][j]; w->qx[k][i] = w->gX[k][i]; w->Hxx[k][i][i] += prox; }
        }
        if (k < N) {
            for (int i = 0; i < nu; i++) { for (int j = 0; j < nu; j++) w->Huu[k][i][j] = w->RW2[i][j]; w->Huu[k][i][i] += prox; w->qu[k][i] = w->gU[k][i]; for (int j = 0; j < nx; j++) w->Hux[k][i][j] = 0; }
            if (use_exact == 2) {
                /* - sum_j lam_{k+1,j} d2 f_j : only f3, f4 (base.py:23-24) are nonlinear */
                double sn = sin(w->X[k][2]), cs = cos(w->X[k][2]), l3 = w->lam[k + 1][3], l4 = w->lam[k + 1][4], dt = c->dt;
                w->Hxx[k][2][2] -= dt * w->U[k][0] * (-l3 * cs - l4 * sn);
                w->Hxx[k][4][5] -= -dt * l3; w->Hxx[k][5][4] -= -dt * l3;
                w->Hxx[k][3][5] -= dt * l4;  w->Hxx[k][5][3] -= dt * l4;
                w->Hux[k][0][2] -= dt * (-l3 * sn + l4 * cs);
            }
        }
        for (int r = 0; r < w->nrow; r++) {
            if (!w->act[k][r]) continue;
            double tt = w->t[k][r], zz = w->z[k][r], wt = zz / tt, rh = w->h[k][r] + tt, zh = mu / tt + wt * rh;
            if (r < 2 * nu) {
                int j = r < nu ? r : r - nu; double sg = r < nu ? -1.0 : 1.0;
                w->Huu[k][j][j] += wt; w->qu[k][j] += sg * zh;
            } else if (r < 2 * nu + 2 * nx) {
                int q = r - 2 * nu; int j = q < nx ? q : q - nx; double sg = q < nx ? -1.0 : 1.0;
                w->Hxx[k][j][j] += wt; w->qx[k][j] += sg * zh;
            } else {
                int ks; double jx[NXM]; memset(jx, 0, sizeof(jx));
                if (r < 2 * nu + 2 * nx + w->M) {
                    int m = r - 2 * nu - 2 * nx; ks = k;
                    jx[0] = w->gcirc[k][m][0]; jx[1] = w->gcirc[k][m][1];
                    if (use_exact) {
                        w->Hxx[k][0][0] += zz * w->hcirc[k][m][0]; w->Hxx[k][0][1] += zz * w->hcirc[k][m][1];
                        w->Hxx[k][1][0] += zz * w->hcirc[k][m][1]; w->Hxx[k][1][1] += zz * w->hcirc[k][m][2];
                    }
                } else if (r < 2 * nu + 2 * nx + w->M + 4 || c->kind != 0) {
                    int i = r - 2 * nu - 2 * nx - w->M; ks = slack_idx(w, k);
                    for (int j = 0; j < 6; j++) jx[YIDX[j]] = w->gself[k][i][j];
                } else if (r >= SL_Q8(w, 0)) {
                    continue;   /* rows of the NLP as written: second sweep below */
                } else {
                    int i = r - 2 * nu - 2 * nx - w->M - 4; ks = k;     /* half-space rows: s[k] (s[N] at the end, :268) */
                    for (int j = 0; j < 6; j++) jx[YIDX[j]] = w->ghs[k][i][j];
                    if (use_exact)   /* curvature of n.FK point over (psi, q1, q2, q3) = y-indices 2..5 */
                        for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) {
                            int hi = a >= b ? a : b, lo = a >= b ? b : a;
                            w->Hxx[k][YIDX[2 + a]][YIDX[2 + b]] += zz * w->hhs[k][i][hi * (hi + 1) / 2 + lo];
                        }
                }
                for (int i = 0; i < nx; i++) { for (int j = 0; j < nx; j++) w->Hxx[k][i][j] += wt * jx[i] * jx[j]; w->qx[k][i] += jx[i] * zh; }
                w->hss[ks] += wt; w->gss[ks] -= zh;
                if (ks == k) for (int i = 0; i < nx; i++) w->vx[k][i] += wt * jx[i];
                else for (int i = 0; i < nx; i++) w->vxN[i] += wt * jx[i];
            }
        }
    }
    for (int k = 1; k <= N && w->nq8; k++)
        for (int e = 0; e < w->nq8; e++) {
            const int r = SL_Q8(w, e), br = w->q8br[k][e], kk = k - br;
            double tt = w->t[k][r], zz = w->z[k][r], wt = zz / tt, rh = w->h[k][r] + tt, zh = mu / tt + wt * rh;
            double jx[NXM]; memset(jx, 0, sizeof(jx));
            for (int j = 0; j < 6; j++) jx[YIDX[j]] = w->gq8[k][e][j];
            if (use_exact)
                for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) {
                    int hi = a >= b ? a : b, lo = a >= b ? b : a;
                    w->Hxx[kk][YIDX[2 + a]][YIDX[2 + b]] += zz * w->hq8[k][e][hi * (hi + 1) / 2 + lo];
                }
            for (int i = 0; i < nx; i++) { for (int j = 0; j < nx; j++) w->Hxx[kk][i][j] += wt * jx[i] * jx[j]; w->qx[kk][i] += jx[i] * zh; }
            w->hss[k] += wt; w->gss[k] -= zh;
            if (br) for (int i = 0; i < nx; i++) w->vq[k][i] += wt * jx[i];      /* couples s_k with x_{k-1} */
            else for (int i = 0; i < nx; i++) w->vx[k][i] += wt * jx[i];
        }
    /* Schur complement of s_k (H_xs = -v).  s_k is tied to x_k by its own rows (v), for k = N-1 also to x_N by the terminal
     * self rows (vN, quirk Q1), and in the NLP as written to x_{k-1} by the rows whose previous-stage entry attains the max
     * (vq).  A slack that reaches back is eliminated one stage earlier, everything expressed in (x_{k-1}, u_{k-1}) through
     * dx_k = A dx_{k-1} + B du_{k-1} + c:  a = vq + A^T v, b = B^T v, gamma = g_s - v.c.  s_{N-1} reaching back touches three
     * stages (x_{N-2}, x_{N-1}, x_N): it is not eliminated at all but kept as a border variable sigma of the stage-wise system
     * (w->sig): [K -v~; -v~^T h] [d; dsigma] = -[q; g_s], solved with one extra right-hand side through the same recursion
     * (column 2 of pvv / kfv, like the two multipliers of the terminal equality).  (Round 2 kept only the diagonal block of
     * x_{N-2}: an inexact Newton matrix whose iteration stalled at a dual residual of 1e-2..1e-6 on 2 % of the starts.) */
    w->sig = 0;
    if (w->nq8 && N >= 2) for (int i = 0; i < nx; i++) if (w->vq[N - 1][i] != 0.0) w->sig = 1;
    for (int k = 0; k <= N; k++) {
        int back = 0;
        if (w->nq8 && k >= 1 && k != N - 1) for (int i = 0; i < nx; i++) if (w->vq[k][i] != 0.0) back = 1;
        double ih = 1.0 / w->hss[k];
        if (back) {
            const int km = k - 1;
            double a[NXM], b[NUM], gam = w->gss[k];
            for (int i = 0; i < nx; i++) { a[i] = w->vq[k][i]; for (int j = 0; j < nx; j++) a[i] += w->A[km][j][i] * w->vx[k][j]; }
            for (int i = 0; i < nu; i++) { b[i] = 0; for (int j = 0; j < nx; j++) b[i] += w->B[km][j][i] * w->vx[k][j]; }
            for (int j = 0; j < nx; j++) gam -= w->vx[k][j] * w->c[km][j];
            for (int i = 0; i < nx; i++) { for (int j = 0; j < nx; j++) w->Hxx[km][i][j] -= a[i] * a[j] * ih; w->qx[km][i] += a[i] * gam * ih; }
            for (int i = 0; i < nu; i++) {
                for (int j = 0; j < nx; j++) w->Hux[km][i][j] -= b[i] * a[j] * ih;
                for (int j = 0; j < nu; j++) w->Huu[km][i][j] -= b[i] * b[j] * ih;
                w->qu[km][i] += b[i] * gam * ih;
            }
            continue;
        }
        if (k == N - 1 && w->sig) continue;   /* border variable: no elimination */
        double a[NXM], b[NUM], gam = w->gss[k];
        for (int i = 0; i < nx; i++) a[i] = w->vx[k][i];
        for (int i = 0; i < nu; i++) b[i] = 0;
        if (k == N - 1) {
            for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) a[i] += w->A[k][j][i] * w->vxN[j];
            for (int i = 0; i < nu; i++) for (int j = 0; j < nx; j++) b[i] += w->B[k][j][i] * w->vxN[j];
            for (int j = 0; j < nx; j++) gam -= w->vxN[j] * w->c[k][j];
        }
        for (int i = 0; i < nx; i++) { for (int j = 0; j < nx; j++) w->Hxx[k][i][j] -= a[i] * a[j] * ih; w->qx[k][i] += a[i] * gam * ih; }
        if (k == N - 1)
            for (int i = 0; i < nu; i++) {
                for (int j = 0; j < nx; j++) w->Hux[k][i][j] -= b[i] * a[j] * ih;
                for (int j = 0; j < nu; j++) w->Huu[k][i][j] -= b[i] * b[j] * ih;
                w->qu[k][i] += b[i] * gam * ih;
            }
    }
    /* Riccati: P_k, p_k overwrite Hxx[k], qx[k] */
    if (c->terminal_xy_eq)
        for (int j = 0; j < 2; j++) {   /* augmentation rho/2 |E dx_N - e|^2 (see ipm_numpy.py: same Newton step, better-posed recursion) */
            w->Hxx[N][j][j] += RHO_EQ;
            w->qx[N][j] -= RHO_EQ * (w->xref[N * nx + j] - w->X[N][j]);
        }
    memset(w->pvv[N], 0, sizeof(w->pvv[N]));
    w->pvv[N][0][0] = 1.0; w->pvv[N][1][1] = 1.0; /* E^T, E = [I2 0] */
    if (w->sig) for (int i = 0; i < nx; i++) w->pvv[N][i][2] = -w->vxN[i];   /* "gradient" of the border column: -v~ */
    for (int k = N - 1; k >= 0; k--) {
        double (*P)[NXM] = w->Hxx[k + 1]; double *p = w->qx[k + 1];
        double PA[NXM][NXM], PB[NXM][NUM], pc[NXM];
        for (int i = 0; i < nx; i++) {
            for (int j = 0; j < nx; j++) { double v = 0; for (int l = 0; l < nx; l++) v += P[i][l] * w->A[k][l][j]; PA[i][j] = v; }
            for (int j = 0; j < nu; j++) { double v = 0; for (int l = 0; l < nx; l++) v += P[i][l] * w->B[k][l][j]; PB[i][j] = v; }
            double v = p[i]; for (int l = 0; l < nx; l++) v += P[i][l] * w->c[k][l]; pc[i] = v;
        }
        double F[NXM][NXM], G[NUM][NXM], Hh[NUM][NUM], gx[NXM], gu[NUM];
        for (int i = 0; i < nx; i++) {
            for (int j = 0; j < nx; j++) { double v = w->Hxx[k][i][j]; for (int l = 0; l < nx; l++) v += w->A[k][l][i] * PA[l][j]; F[i][j] = v; }
            double v = w->qx[k][i]; for (int l = 0; l < nx; l++) v += w->A[k][l][i] * pc[l]; gx[i] = v;
        }
        for (int i = 0; i < nu; i++) {
            for (int j = 0; j < nx; j++) { double v = w->Hux[k][i][j]; for (int l = 0; l < nx; l++) v += w->B[k][l][i] * PA[l][j]; G[i][j] = v; }
            for (int j = 0; j < nu; j++) { double v = w->Huu[k][i][j]; for (int l = 0; l < nx; l++) v += w->B[k][l][i] * PB[l][j]; Hh[i][j] = v; }
            double v = w->qu[k][i]; for (int l = 0; l < nx; l++) v += w->B[k][l][i] * pc[l]; gu[i] = v;
        }
        double L[NUM][NUM];
        if (!chol(Hh, nu, L)) return 0;
        for (int j = 0; j < nx; j++) {
            double col[NUM]; for (int i = 0; i < nu; i++) col[i] = G[i][j];
            chol_solve(L, nu, col);
            for (int i = 0; i < nu; i++) w->K[k][i][j] = -col[i];
        }
        { double col[NUM]; for (int i = 0; i < nu; i++) col[i] = gu[i]; chol_solve(L, nu, col); for (int i = 0; i < nu; i++) w->kf[k][i] = -col[i]; }
        for (int cc = 0; cc < 3; cc++) {   /* sensitivities of the direction w.r.t. the two terminal multipliers / the border slack */
                if (cc < 2 ? !c->terminal_xy_eq : !w->sig) continue;
                double col[NUM];
                for (int i = 0; i < nu; i++) { double v = 0; for (int l = 0; l < nx; l++) v += w->B[k][l][i] * w->pvv[k + 1][l][cc]; col[i] = v; }
                chol_solve(L, nu, col);
                for (int i = 0; i < nu; i++) w->kfv[k][i][cc] = -col[i];
                for (int i = 0; i < nx; i++) {
                    double v = 0;
                    for (int l = 0; l < nx; l++) v += w->A[k][l][i] * w->pvv[k + 1][l][cc];
                    for (int l = 0; l < nu; l++) v += G[l][i] * w->kfv[k][l][cc];
                    if (cc == 2 && k == N - 1) v -= w->vx[N - 1][i];
                    if (cc == 2 && k == N - 2) v -= w->vq[N - 1][i];
                    w->pvv[k][i][cc] = v;
                }
            }
        double Pn[NXM][NXM];
        for (int i = 0; i < nx; i++) {
            for (int j = 0; j < nx; j++) { double v = F[i][j]; for (int l = 0; l < nu; l++) v += G[l][i] * w->K[k][l][j]; Pn[i][j] = v; }
            double v = gx[i]; for (int l = 0; l < nu; l++) v += G[l][i] * w->kf[k][l]; w->qx[k][i] = v;
        }
        for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) w->Hxx[k][i][j] = 0.5 * (Pn[i][j] + Pn[j][i]);
    }
    if (w->sig) {
        /* K d1 = v~ (homogeneous dynamics): roll-out of the border column; h - v~.d1 is the pivot of the border variable */
        memset(w->dxs, 0, sizeof(w->dxs));
        for (int k = 0; k < N; k++) {
            double du[NUM];
            for (int i = 0; i < nu; i++) { double v = w->kfv[k][i][2]; for (int j = 0; j < nx; j++) v += w->K[k][i][j] * w->dxs[k][j]; du[i] = v; }
            for (int i = 0; i < nx; i++) {
                double v = 0;
                for (int j = 0; j < nx; j++) v += w->A[k][i][j] * w->dxs[k][j];
                for (int j = 0; j < nu; j++) v += w->B[k][i][j] * du[j];
                w->dxs[k + 1][i] = v;
            }
        }
        double t1 = 0;
        for (int j = 0; j < nx; j++) t1 += w->vq[N - 1][j] * w->dxs[N - 2][j] + w->vx[N - 1][j] * w->dxs[N - 1][j] + w->vxN[j] * w->dxs[N][j];
        w->sig_den = w->hss[N - 1] - t1;
        if (!(w->sig_den > 0.0)) return 0;
    }
    return 1;
}

/* Newton direction from the factorisation in w (gains, cost-to-go): border solve, roll-out, multiplier and row steps,
 * fraction to the boundary; outputs the primal / dual step bounds and the directional derivative of the barrier objective */
static void direction(work *w, double mu, double *ap_out, double *ad_out, double *dphi_out) {
    const oracle_cfg *cfg = w->cfg;
    const int nx = w->nx, nu = w->nu, N = w->N;
    w->nu_new[0] = w->nu_new[1] = 0; w->dsig = 0;
    if (cfg->terminal_xy_eq || w->sig) {
        /* the direction is affine in the border variables y = (nu0, nu1, dsigma): roll out the y = 0 solution and the
         * sensitivities (same gains), then the small system
         *   E (dx_N + D y) = e                        (terminal equality)
         *   h dsigma - v~.(d + D y) = -g_s            (stationarity in s_{N-1}) */
        static __thread double d0[NSM][NXM], Dv[NSM][NXM][2];
        memset(d0, 0, sizeof(d0)); memset(Dv, 0, sizeof(Dv));
        for (int k = 0; k < N; k++) {
            double u0[NUM], Uv[NUM][2];
            for (int i = 0; i < nu; i++) {
                double v = w->kf[k][i], v0 = w->kfv[k][i][0], v1 = w->kfv[k][i][1];
                for (int j = 0; j < nx; j++) { v += w->K[k][i][j] * d0[k][j]; v0 += w->K[k][i][j] * Dv[k][j][0]; v1 += w->K[k][i][j] * Dv[k][j][1]; }
                u0[i] = v; Uv[i][0] = v0; Uv[i][1] = v1;
            }
            for (int i = 0; i < nx; i++) {
                double v = w->c[k][i], v0 = 0, v1 = 0;
                for (int j = 0; j < nx; j++) { v += w->A[k][i][j] * d0[k][j]; v0 += w->A[k][i][j] * Dv[k][j][0]; v1 += w->A[k][i][j] * Dv[k][j][1]; }
                for (int j = 0; j < nu; j++) { v += w->B[k][i][j] * u0[j]; v0 += w->B[k][i][j] * Uv[j][0]; v1 += w->B[k][i][j] * Uv[j][1]; }
                d0[k + 1][i] = v; Dv[k + 1][i][0] = v0; Dv[k + 1][i][1] = v1;
            }
        }
        double Ms[3][4]; int act3[3] = {cfg->terminal_xy_eq, cfg->terminal_xy_eq, w->sig};
        memset(Ms, 0, sizeof(Ms));
        for (int r = 0; r < 2; r++) {
            Ms[r][0] = Dv[N][r][0]; Ms[r][1] = Dv[N][r][1]; Ms[r][2] = w->sig ? w->dxs[N][r] : 0.0;
            Ms[r][3] = w->xref[N * nx + r] - w->X[N][r] - d0[N][r];
        }
        if (w->sig) {
            double t0 = 0, tv[2] = {0, 0};
            for (int j = 0; j < nx; j++) {
                t0 += w->vq[N - 1][j] * d0[N - 2][j] + w->vx[N - 1][j] * d0[N - 1][j] + w->vxN[j] * d0[N][j];
                for (int cc = 0; cc < 2; cc++) tv[cc] += w->vq[N - 1][j] * Dv[N - 2][j][cc] + w->vx[N - 1][j] * Dv[N - 1][j][cc] + w->vxN[j] * Dv[N][j][cc];
            }
            Ms[2][0] = -tv[0]; Ms[2][1] = -tv[1]; Ms[2][2] = w->sig_den; Ms[2][3] = -w->gss[N - 1] + t0;
        }
        for (int r = 0; r < 3; r++) if (!act3[r]) { for (int q = 0; q < 4; q++) Ms[r][q] = 0.0; for (int q = 0; q < 3; q++) Ms[q][r] = 0.0; Ms[r][r] = 1.0; }
        for (int pcol = 0; pcol < 3; pcol++) {   /* Gaussian elimination with row pivoting (3 x 3) */
            int pr = pcol; for (int r = pcol + 1; r < 3; r++) if (fabs(Ms[r][pcol]) > fabs(Ms[pr][pcol])) pr = r;
            if (pr != pcol) for (int q = 0; q < 4; q++) { double tq = Ms[pr][q]; Ms[pr][q] = Ms[pcol][q]; Ms[pcol][q] = tq; }
            for (int r = 0; r < 3; r++) if (r != pcol) { double f = Ms[r][pcol] / Ms[pcol][pcol]; for (int q = pcol; q < 4; q++) Ms[r][q] -= f * Ms[pcol][q]; }
        }
        double y[3]; for (int r = 0; r < 3; r++) y[r] = Ms[r][3] / Ms[r][r];
        w->nu_new[0] = y[0]; w->nu_new[1] = y[1]; w->dsig = y[2];
        for (int k = 0; k < N; k++) for (int i = 0; i < nu; i++) {
            if (cfg->terminal_xy_eq) w->kf[k][i] += w->kfv[k][i][0] * y[0] + w->kfv[k][i][1] * y[1];
            if (w->sig) w->kf[k][i] += w->kfv[k][i][2] * y[2];
        }
    }
    for (int j = 0; j < nx; j++) w->dX[0][j] = 0;
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < nu; i++) { double v = w->kf[k][i]; for (int j = 0; j < nx; j++) v += w->K[k][i][j] * w->dX[k][j]; w->dU[k][i] = v; }
        for (int i = 0; i < nx; i++) {
            double v = w->c[k][i];
            for (int j = 0; j < nx; j++) v += w->A[k][i][j] * w->dX[k][j];
            for (int j = 0; j < nu; j++) v += w->B[k][i][j] * w->dU[k][j];
            w->dX[k + 1][i] = v;
        }
    }
    for (int k = 1; k <= N; k++) for (int i = 0; i < nx; i++) {
        double v = w->qx[k][i]; for (int j = 0; j < nx; j++) v += w->Hxx[k][i][j] * w->dX[k][j];
        if (cfg->terminal_xy_eq) v += w->pvv[k][i][0] * w->nu_new[0] + w->pvv[k][i][1] * w->nu_new[1];
        if (w->sig) v += w->pvv[k][i][2] * w->dsig;
        w->lamn[k][i] = -v; }
    for (int k = 0; k <= N; k++) {
        double vdx = 0; for (int j = 0; j < nx; j++) vdx += w->vx[k][j] * w->dX[k][j];
        if (k == N - 1) for (int j = 0; j < nx; j++) vdx += w->vxN[j] * w->dX[N][j];
        if (w->nq8 && k >= 1) for (int j = 0; j < nx; j++) vdx += w->vq[k][j] * w->dX[k - 1][j];
        w->ds[k] = -(w->gss[k] - vdx) / w->hss[k];
    }
    if (w->nq8)   /* a slack eliminated one stage earlier is not part of its own stage's cost-to-go: its pull on x_k enters the
                   * multiplier of the dynamics directly,  lam_k+ = -(P_k dx_k + p_k) + v_k ds_k */
        for (int k = 1; k <= N; k++) {
            int back = 0;
            if (k != N - 1) for (int i = 0; i < nx; i++) if (w->vq[k][i] != 0.0) back = 1;
            if (back) for (int i = 0; i < nx; i++) w->lamn[k][i] += w->vx[k][i] * w->ds[k];
        }
    double tau = fmax(0.99, 1 - mu), ap = 1, ad = 1, dphi = 0;
    for (int k = 0; k <= N; k++) {
        for (int r = 0; r < w->nrow; r++) {
            if (!w->act[k][r]) continue;
            double jd;
            if (r < 2 * nu) { int j = r < nu ? r : r - nu; jd = (r < nu ? -1.0 : 1.0) * w->dU[k][j]; }
            else if (r < 2 * nu + 2 * nx) { int q = r - 2 * nu; int j = q < nx ? q : q - nx; jd = (q < nx ? -1.0 : 1.0) * w->dX[k][j]; }
            else if (r < 2 * nu + 2 * nx + w->M) { int m = r - 2 * nu - 2 * nx; jd = w->gcirc[k][m][0] * w->dX[k][0] + w->gcirc[k][m][1] * w->dX[k][1] - w->ds[k]; }
            else if (r < 2 * nu + 2 * nx + w->M + 4) { int i = r - 2 * nu - 2 * nx - w->M; jd = -w->ds[slack_idx(w, k)]; for (int j = 0; j < 6; j++) jd += w->gself[k][i][j] * w->dX[k][YIDX[j]]; }
            else if (r < SL_Q8(w, 0)) { int i = r - 2 * nu - 2 * nx - w->M - 4; jd = -w->ds[k]; for (int j = 0; j < 6; j++) jd += w->ghs[k][i][j] * w->dX[k][YIDX[j]]; }
            else { int e = r - SL_Q8(w, 0), kk = k - w->q8br[k][e]; jd = -w->ds[k]; for (int j = 0; j < 6; j++) jd += w->gq8[k][e][j] * w->dX[kk][YIDX[j]]; }
            double tt = w->t[k][r], zz = w->z[k][r];
            double dtv = -(w->h[k][r] + tt) - jd, dzv = mu / tt - zz - (zz / tt) * dtv;
            w->dt_[k][r] = dtv; w->dz[k][r] = dzv;
            if (dtv < 0) { double a = -tau * tt / dtv; if (a < ap) { ap = a; w->dbg_k = k; w->dbg_r = r; w->dbg_t = tt; w->dbg_dt = dtv; w->dbg_z = zz; } }
            if (dzv < 0) { double a = -tau * zz / dzv; if (a < ad) ad = a; }
            dphi -= mu * dtv / tt;
        }
        for (int j = 0; j < nx; j++) dphi += w->gX[k][j] * w->dX[k][j];
        if (k < N) for (int j = 0; j < nu; j++) dphi += w->gU[k][j] * w->dU[k][j];
        dphi += w->gs[k] * w->ds[k];
    }
    *ap_out = ap; *ad_out = ad; *dphi_out = dphi;
}

typedef struct { double th, phi; } fent;


/*
 * Solve one instance.  Inputs row-major: x_init[nx] (already clipped by the caller for the
 * whole-body kind, mpc_wholebody_qref.py:290-291), traj_ref[(N+1)*nx], u_ref[N*nu],
 * u_last[N*nu] (U_last parameter AND initial U, :303,:310), X0 (NULL => tile(x_init), :302; base
 * kind passes the previous X, mpc_base.py:200), obs.  Outputs X,U,s; returns status
 * (0 converged, 1 max_iter, 2 numerical failure).
 */
int mmpc_oracle_solve(const oracle_cfg *cfg, const double *x_init, const double *traj_ref, const double *u_ref,
                      const double *u_last, const double *X0, const double *U0, const double *obs, double *Xo, double *Uo,
                      double *so, int *iters_out, double *cost_out, double *err_out) {
    work *w = (work *)calloc(1, sizeof(work));
    if (!w) return 2;
    w->cfg = cfg; w->N = cfg->N; w->M = cfg->M;
    w->nx = cfg->kind == 0 ? 9 : 6; w->nu = cfg->kind == 0 ? 5 : 2;
    int nx = w->nx, nu = w->nu, N = w->N;
    w->nhs = (cfg->kind == 0 && !cfg->pose_ref && cfg->L > 0) ? 6 : 0;
    w->nq8 = (w->nhs && cfg->as_written && cfg->L >= 2) ? 6 * (cfg->L - 1) : 0;
    w->nrow = 2 * nu + 2 * nx + w->M + ((cfg->kind == 0 && !cfg->pose_ref) ? 4 : 0) + w->nhs + w->nq8;
    w->xinit = x_init; w->xref = traj_ref; w->uref = u_ref; w->ulast = u_last; w->obs = obs;
    for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) {
        w->Q2[i][j] = cfg->Q[i * nx + j] + cfg->Q[j * nx + i]; w->P2[i][j] = cfg->P[i * nx + j] + cfg->P[j * nx + i]; }
    for (int i = 0; i < nu; i++) for (int j = 0; j < nu; j++)
        w->RW2[i][j] = cfg->R[i * nu + j] + cfg->R[j * nu + i] + cfg->W[i * nu + j] + cfg->W[j * nu + i];
    for (int k = 0; k <= N; k++) {
        for (int j = 0; j < nx; j++) w->X[k][j] = (X0 && k > 0) ? X0[k * nx + j] : x_init[j];
        if (k < N) for (int j = 0; j < nu; j++) w->U[k][j] = U0 ? U0[k * nu + j] : u_last[k * nu + j];
        w->s[k] = 0;
    }
    setup_rows(w);
    /* bound push of the initial point (see oracle/ipm_numpy.py): finite simple bounds are kept >= BOUND_PUSH away */
    for (int k = 0; k <= N; k++) {
        if (k < N) for (int j = 0; j < nu; j++)
            w->U[k][j] = bound_push(w->U[k][j], w->act[k][SL_ULO(w, j)] ? w->bnd[k][SL_ULO(w, j)] : -INFINITY,
                                    w->act[k][SL_UHI(w, j)] ? w->bnd[k][SL_UHI(w, j)] : INFINITY);
        if (k >= 1) for (int j = 0; j < nx; j++)
            w->X[k][j] = bound_push(w->X[k][j], w->act[k][SL_XLO(w, j)] ? w->bnd[k][SL_XLO(w, j)] : -INFINITY,
                                    w->act[k][SL_XHI(w, j)] ? w->bnd[k][SL_XHI(w, j)] : INFINITY);
    }
    double mu = cfg->mu_init;
    eval_rows(w, w->X, w->U, w->s, w->h, 0);
    int nrows_act = 0;
    for (int k = 0; k <= N; k++) for (int r = 0; r < w->nrow; r++) if (w->act[k][r]) {
        double v = -w->h[k][r]; w->t[k][r] = v > 1e-2 ? v : 1e-2; w->z[k][r] = mu / w->t[k][r]; nrows_act++; }
    int status = 1, it = 0, nf = 0, nsmall = 0;
    double prox = 0.0, delta_last = 0.0, delta_prev_it = 0.0; work *sv = 0;
    double th_max = 0, th_min = 0, E0 = 0;
    fent filt[FCAP];
    int filt_init = 0, nfilt = 0;
    const double tol = cfg->tol;
    static __thread double Xn[NSM][NXM], Un[NSM][NUM], sn[NSM], tn[NSM][RMX];
    for (it = 0; it <= cfg->max_iter; it++) {
        /* ---- multiplier safeguard (IPOPT eq. 16): z within [mu/(kappa t), kappa mu/t] */
        for (int k = 0; k <= N; k++) for (int r = 0; r < w->nrow; r++) if (w->act[k][r]) {
            double lo_ = mu / (KAPPA_SIGMA * w->t[k][r]), hi_ = KAPPA_SIGMA * mu / w->t[k][r];
            if (w->z[k][r] < lo_) w->z[k][r] = lo_; else if (w->z[k][r] > hi_) w->z[k][r] = hi_;
        }
        /* ---- evaluation at the current point */
        eval_rows(w, w->X, w->U, w->s, w->h, 1);
        for (int k = 0; k <= N; k++) {
            state_cost(w, k, w->X[k], w->gX[k], 0, 0);
            w->gs[k] = 2 * cfg->S * w->s[k];
            if (k < N) {
                for (int i = 0; i < nu; i++) {
                    double v = 0;
                    for (int j = 0; j < nu; j++)
                        v += (cfg->R[i * nu + j] + cfg->R[j * nu + i]) * (w->U[k][j] - w->uref[k * nu + j])
                           + (cfg->W[i * nu + j] + cfg->W[j * nu + i]) * (w->U[k][j] - w->ulast[k * nu + j]);
                    w->gU[k][i] = v;
                }
                f_jac(cfg->kind, cfg->dt, w->X[k], w->U[k], w->A[k], w->B[k]);
                double xn[NXM]; f_dyn(cfg->kind, cfg->dt, w->X[k], w->U[k], xn);
                for (int j = 0; j < nx; j++) w->c[k][j] = xn[j] - w->X[k + 1][j];
            }
        }
        /* ---- KKT error */
        double err_d = 0, err_p = 0, comp0 = 0, compmu = 0, zsum = 0, zrow = 0;
        {
            static __thread double rdx[NSM][NXM], rdu[NSM][NUM], rds[NSM];
            for (int k = 0; k <= N; k++) { memcpy(rdx[k], w->gX[k], sizeof(rdx[k])); memcpy(rdu[k], w->gU[k], sizeof(rdu[k])); rds[k] = w->gs[k]; }
            for (int k = 0; k <= N; k++) for (int r = 0; r < w->nrow; r++) {
                if (!w->act[k][r]) continue;
                double zz = w->z[k][r], tt = w->t[k][r];
                zsum += zz; zrow += zz;
                double cp = tt * zz; if (cp > comp0) comp0 = cp; if (fabs(cp - mu) > compmu) compmu = fabs(cp - mu);
                double rh = fabs(w->h[k][r] + tt); if (rh > err_p) err_p = rh;
                if (r < 2 * nu) { int j = r < nu ? r : r - nu; rdu[k][j] += (r < nu ? -zz : zz); }
                else if (r < 2 * nu + 2 * nx) { int q = r - 2 * nu; int j = q < nx ? q : q - nx; rdx[k][j] += (q < nx ? -zz : zz); }
                else if (r < 2 * nu + 2 * nx + w->M) { int m = r - 2 * nu - 2 * nx; rdx[k][0] += w->gcirc[k][m][0] * zz; rdx[k][1] += w->gcirc[k][m][1] * zz; rds[k] -= zz; }
                else if (r < 2 * nu + 2 * nx + w->M + 4) { int i = r - 2 * nu - 2 * nx - w->M; for (int j = 0; j < 6; j++) rdx[k][YIDX[j]] += w->gself[k][i][j] * zz; rds[slack_idx(w, k)] -= zz; }
                else if (r < 2 * nu + 2 * nx + w->M + 4 + w->nhs) { int i = r - 2 * nu - 2 * nx - w->M - 4; for (int j = 0; j < 6; j++) rdx[k][YIDX[j]] += w->ghs[k][i][j] * zz; rds[k] -= zz; }
                else { int e = r - SL_Q8(w, 0), kk = k - w->q8br[k][e]; for (int j = 0; j < 6; j++) rdx[kk][YIDX[j]] += w->gq8[k][e][j] * zz; rds[k] -= zz; }
            }
            for (int k = 0; k < N; k++) {
                for (int i = 0; i < nx; i++) {
                    rdx[k + 1][i] += w->lam[k + 1][i];
                    double v = 0; for (int j = 0; j < nx; j++) v += w->A[k][j][i] * w->lam[k + 1][j]; rdx[k][i] -= v;
                }
                for (int i = 0; i < nu; i++) { double v = 0; for (int j = 0; j < nx; j++) v += w->B[k][j][i] * w->lam[k + 1][j]; rdu[k][i] -= v; }
                for (int j = 0; j < nx; j++) { if (fabs(w->c[k][j]) > err_p) err_p = fabs(w->c[k][j]); zsum += fabs(w->lam[k + 1][j]); }
            }
            if (cfg->terminal_xy_eq)
                for (int j = 0; j < 2; j++) {
                    rdx[N][j] += w->nu_eq[j];
                    double ce = fabs(w->X[N][j] - w->xref[N * nx + j]); if (ce > err_p) err_p = ce;
                    zsum += fabs(w->nu_eq[j]);
                }
            for (int k = 0; k <= N; k++) {
                if (k > 0) for (int i = 0; i < nx; i++) if (fabs(rdx[k][i]) > err_d) { err_d = fabs(rdx[k][i]); if (getenv("MMPC_ORACLE_DEBUG2")) fprintf(stderr, "  rdx[%d][%d] %.3e\n", k, i, rdx[k][i]); }
                if (k < N) for (int i = 0; i < nu; i++) if (fabs(rdu[k][i]) > err_d) { err_d = fabs(rdu[k][i]); if (getenv("MMPC_ORACLE_DEBUG2")) fprintf(stderr, "  rdu[%d][%d] %.3e\n", k, i, rdu[k][i]); }
                if (fabs(rds[k]) > err_d) { err_d = fabs(rds[k]); if (getenv("MMPC_ORACLE_DEBUG2")) fprintf(stderr, "  rds[%d] %.3e\n", k, rds[k]); }
            }
        }
        double sd = zsum / (nrows_act + (N + 1) * nx); sd = (sd > 100.0 ? sd : 100.0) / 100.0;
        /* IPOPT's scaling of the complementarity (Waechter & Biegler 2006, eq. 5-6): s_c from the bound multipliers alone */
        double sc = zrow / (nrows_act > 0 ? nrows_act : 1); sc = (sc > 100.0 ? sc : 100.0) / 100.0;
        if (LAB[0] < 1.0) sc = sd;
        E0 = fmax(fmax(err_d / sd, err_p), comp0 / sc);
        double Emu = fmax(fmax(err_d / sd, err_p), compmu / sc);
        if (!(E0 == E0)) { status = 2; break; }
        if (E0 <= tol) { status = 0; break; }
        if (it == cfg->max_iter) break;
        int changed = 0;
        while (Emu <= 10.0 * mu && mu > tol / 10) {
            double m1 = 0.2 * mu, m2 = pow(mu, 1.5);
            mu = fmax(tol / 10, fmin(m1, m2));
            compmu = 0;
            for (int k = 0; k <= N; k++) for (int r = 0; r < w->nrow; r++) if (w->act[k][r]) { double d = fabs(w->t[k][r] * w->z[k][r] - mu); if (d > compmu) compmu = d; }
            Emu = fmax(fmax(err_d / sd, err_p), compmu / sc);
            changed = 1;
        }
        if (changed) filt_init = 0;
        /* ---- Newton direction */
        int rung = 2; double delta_used = 0.0;
#define NEWTON_SOLVE(FAILSTMT)                                                                                              \
        rung = 2; delta_used = 0.0;                                                                                          \
        if (LAB[1] >= 1.0 && !cfg->terminal_xy_eq) {                                                                         \
            /* IPOPT's inertia correction (Waechter & Biegler 2006, Algorithm IC): H + delta_w I with the smallest delta_w  \
             * of a geometric sequence for which every pivot of the recursion is positive.  Not with the terminal-xy        \
             * equality: the multipliers of the regularised system grow like delta_w there (the full correction             \
             * E dx_N = e is forced whatever the damping) and feed back into lam^T d2f - Gauss-Newton as in round 3 */     \
            int okf = (LAB[25] > 0 && delta_prev_it > LAB[25]) ? 0 : factor(w, mu, 2, prox);                                 \
            if (!okf) {                                                                                                      \
                const double d0 = LAB[6] > 0 ? LAB[6] : 1e-4, kfirst = LAB[7] > 0 ? LAB[7] : 100.0, kup = LAB[8] > 0 ? LAB[8] : 8.0, kdn = LAB[9] > 0 ? LAB[9] : 1.0 / 3.0; \
                double dw = delta_last == 0.0 ? d0 : fmax(1e-20, kdn * delta_last);                                          \
                while (!(okf = factor(w, mu, 2, prox + dw))) {                                                               \
                    dw *= delta_last == 0.0 ? kfirst : kup;                                                                  \
                    if (dw > 1e40) break;                                                                                    \
                }                                                                                                            \
                if (!okf) { FAILSTMT; }                                                                                      \
                delta_last = dw; delta_used = dw;                                                                            \
            }                                                                                                                \
        } else if (!factor(w, mu, 2, prox)) if ((rung = 1, cfg->terminal_xy_eq || w->nhs > 0 || !factor(w, mu, 1, prox))) { \
            rung = 0;                                                                                                        \
            int okf = factor(w, mu, 0, prox);                                                                                \
            while (!okf && !cfg->terminal_xy_eq && prox < PROX_MAX) {                                                        \
                prox = prox * 4.0 > PROX0 ? prox * 4.0 : PROX0; if (prox > PROX_MAX) prox = PROX_MAX;                        \
                okf = factor(w, mu, 0, prox);                                                                                \
            }                                                                                                                \
            if (!okf) { FAILSTMT; }                                                                                          \
        }
        int nfail = 0;
        NEWTON_SOLVE(nfail = 1)
        if (nfail) { status = 2; break; }
        delta_prev_it = delta_used;
        double ap = 1, ad = 1, dphi = 0;
        direction(w, mu, &ap, &ad, &dphi);
        /* ---- filter line search */
        double phi0, th0;
        merit_parts(w, w->X, w->U, w->s, w->t, mu, &phi0, &th0);
        if (!filt_init) { nfilt = 0; th_max = 1e4 * fmax(1.0, th0); th_min = 1e-4 * fmax(1.0, th0); filt_init = 1; }
        double alpha = ap; int accepted = 0, soc_used = 0, z_done = 0;
        const int soc_max = (int)LAB[2];
#define TRIAL_POINT(A)                                                                                                      \
        for (int k = 0; k <= N; k++) {                                                                                       \
            for (int j = 0; j < nx; j++) Xn[k][j] = w->X[k][j] + (A) * w->dX[k][j];                                          \
            if (k < N) for (int j = 0; j < nu; j++) Un[k][j] = w->U[k][j] + (A) * w->dU[k][j];                               \
            sn[k] = w->s[k] + (A) * w->ds[k];                                                                                \
            for (int r = 0; r < w->nrow; r++) if (w->act[k][r]) tn[k][r] = w->t[k][r] + (A) * w->dt_[k][r];                  \
        }
        /* acceptance of a trial (th, phi) against the filter and the current point; a0 = the step length the switching and
         * Armijo conditions are evaluated with (a second-order correction is tested with the length of the step it corrects) */
#define ACCEPT_TEST(TH, PHI, A0, OUT)                                                                                       \
        {                                                                                                                    \
            int okf_ = (TH) < th_max;                                                                                        \
            for (int i = 0; i < nfilt && okf_; i++) if ((TH) >= filt[i].th && (PHI) >= filt[i].phi) okf_ = 0;               \
            const int ftype_ = dphi < 0 && th0 <= th_min && (A0) * pow(-dphi, 2.3) > pow(th0, 1.1);                          \
            (OUT) = 0;                                                                                                       \
            if (okf_) {                                                                                                      \
                if (ftype_) { if ((PHI) <= phi0 + 1e-8 * (A0) * dphi + 1e-14 * fabs(phi0)) (OUT) = 1; }                      \
                else if ((TH) <= (1 - 1e-5) * th0 || (PHI) <= phi0 - 1e-5 * th0) {                                           \
                    (OUT) = 1;                                                                                               \
                    fent ne = {(1 - 1e-5) * th0, phi0 - 1e-5 * th0};                                                         \
                    if (nfilt < FCAP) filt[nfilt++] = ne;                                                                    \
                    else { int im = 0; for (int i = 1; i < FCAP; i++) if (filt[i].th > filt[im].th) im = i; filt[im] = ne; } \
                }                                                                                                            \
            }                                                                                                                \
        }
        for (int lspass = 0; lspass < 2 && !accepted; lspass++) {
            alpha = ap;
            for (int ls = 0; ls < 20; ls++) {
                TRIAL_POINT(alpha)
                double phi, th;
                __atomic_fetch_add(&LAB_NTRIAL, 1, __ATOMIC_RELAXED);
                merit_parts(w, Xn, Un, sn, tn, mu, &phi, &th);
                ACCEPT_TEST(th, phi, alpha, accepted)
                if (accepted) break;
                if (ls == 0 && lspass == 0 && soc_max > 0 && th >= th0 && !cfg->terminal_xy_eq && (LAB[20] < 1.0 || th0 <= LAB[20] * th_min)) {
                    /* Second-order correction (Waechter & Biegler 2006, section 2.4): the first trial was rejected and did not
                     * reduce the infeasibility.  The step is recomputed from x_k with c_soc = alpha c(x_k) + c(x_k + alpha d)
                     * in place of the constraint residuals, at most soc_max times (c_soc <- alpha_soc c_soc + c(x_soc)).
                     * The row multipliers take their step of the uncorrected direction first (they do not depend on alpha;
                     * this is what the kernels, which move to a trial point in place, have in their registers at this point),
                     * so the corrected system is the primal-dual Newton system at (x_k, z+).  A correction that fails leaves
                     * x_k where it is: the iteration ends with the multiplier step alone, the next one runs without. */
                    static __thread double accC[NSM][NXM], accR[NSM][RMX], hs_[NSM][RMX], c0s[NSM][NXM], h0s[NSM][RMX];
                    const double alpha0 = alpha;
                    for (int k = 0; k <= N; k++) for (int r = 0; r < w->nrow; r++) if (w->act[k][r]) {
                        w->z[k][r] += ad * w->dz[k][r];
                        const double lo_ = mu / (KAPPA_SIGMA * tn[k][r]), hi_ = KAPPA_SIGMA * mu / tn[k][r];
                        if (w->z[k][r] < lo_) w->z[k][r] = lo_; else if (w->z[k][r] > hi_) w->z[k][r] = hi_;
                    }
                    z_done = 1;
                    if (!sv) sv = (work *)malloc(sizeof(work));
                    memcpy(sv->dX, w->dX, sizeof(w->dX)); memcpy(sv->dU, w->dU, sizeof(w->dU)); memcpy(sv->ds, w->ds, sizeof(w->ds));
                    memcpy(sv->lamn, w->lamn, sizeof(w->lamn)); memcpy(sv->dt_, w->dt_, sizeof(w->dt_));
                    sv->nu_new[0] = w->nu_new[0]; sv->nu_new[1] = w->nu_new[1]; sv->dsig = w->dsig;
                    memcpy(c0s, w->c, sizeof(c0s)); memcpy(h0s, w->h, sizeof(h0s));
                    for (int k = 0; k <= N; k++) {
                        for (int j = 0; j < nx; j++) accC[k][j] = w->c[k][j];
                        for (int r = 0; r < w->nrow; r++) accR[k][r] = w->act[k][r] ? w->h[k][r] + w->t[k][r] : 0.0;
                    }
                    double a_prev = alpha, th_prev = th; int soc_ok = 0;
                    for (int p = 0; p < soc_max; p++) {
                        __atomic_fetch_add(&LAB_NSOC, 1, __ATOMIC_RELAXED);
                        eval_rows(w, Xn, Un, sn, hs_, 0);
                        for (int k = 0; k <= N; k++) {
                            if (k < N) {
                                double xn_[NXM]; f_dyn(cfg->kind, cfg->dt, Xn[k], Un[k], xn_);
                                for (int j = 0; j < nx; j++) { accC[k][j] = a_prev * accC[k][j] + (xn_[j] - Xn[k + 1][j]); w->c[k][j] = accC[k][j]; }
                            }
                            for (int r = 0; r < w->nrow; r++) if (w->act[k][r]) {
                                accR[k][r] = a_prev * accR[k][r] + (hs_[k][r] + tn[k][r]);
                                w->h[k][r] = accR[k][r] - w->t[k][r];
                            }
                        }
                        int sfail = 0;
                        NEWTON_SOLVE(sfail = 1)
                        if (sfail) break;
                        double ap2, ad2, dphi2;
                        direction(w, mu, &ap2, &ad2, &dphi2);
                        TRIAL_POINT(ap2)
                        double phi2, th2;
                        merit_parts(w, Xn, Un, sn, tn, mu, &phi2, &th2);
                        ACCEPT_TEST(th2, phi2, alpha0, soc_ok)
                        if (soc_ok) { alpha = ap2; soc_used = p + 1; __atomic_fetch_add(&LAB_NSOCOK, 1, __ATOMIC_RELAXED); break; }
                        if (th2 > 0.99 * th_prev) break;
                        a_prev = ap2; th_prev = th2;
                    }
                    memcpy(w->c, c0s, sizeof(c0s)); memcpy(w->h, h0s, sizeof(h0s));
                    if (soc_ok) { accepted = 1; break; }
                    __atomic_fetch_add(&LAB_NSOCFAIL, 1, __ATOMIC_RELAXED);
                    /* the correction failed: the line search goes on along the uncorrected direction */
                    memcpy(w->dX, sv->dX, sizeof(w->dX)); memcpy(w->dU, sv->dU, sizeof(w->dU)); memcpy(w->ds, sv->ds, sizeof(w->ds));
                    memcpy(w->lamn, sv->lamn, sizeof(w->lamn)); memcpy(w->dt_, sv->dt_, sizeof(w->dt_));
                    w->nu_new[0] = sv->nu_new[0]; w->nu_new[1] = sv->nu_new[1]; w->dsig = sv->dsig;
                }
                if (ls < 19) alpha *= 0.5;
            }
            if (accepted || nfilt == 0) break;
            nfilt = 0; /* filter reset heuristic: the filter blocked every trial step */
        }
        nf += !accepted;
        if (getenv("MMPC_ORACLE_DEBUG")) fprintf(stderr, "it %d mu %.2e E0 %.3e err_d %.3e err_p %.3e comp %.3e alpha %.3e ap %.3e ad %.3e acc %d prox %.1e dphi %.3e rung %d dw %.1e th %.2e soc %d | k %d r %d t %.2e dt %.2e z %.2e\n", it, mu, E0, err_d, err_p, comp0, alpha, ap, ad, accepted, prox, dphi, rung, delta_used, th0, soc_used, w->dbg_k, w->dbg_r, w->dbg_t, w->dbg_dt, w->dbg_z);
        /* proximal term for crawling iterations (oracle/ipm_numpy.py: Options.prox*) */
        {
            const double PX0 = LAB[15] > 0 ? LAB[15] : PROX0, PXLO = LAB[17] > 0 ? LAB[17] : PROX_LO; const int PXN = LAB[16] > 0 ? (int)LAB[16] : 2;
            const double a_px = LAB[18] >= 1.0 ? ap : alpha;
            nsmall = a_px < PXLO ? nsmall + 1 : 0;
            if (cfg->terminal_xy_eq || LAB[3] >= 1.0) prox = 0.0;   /* the forced correction E dx_N = e makes nu grow like prox */
            else if (a_px < PXLO && (nsmall >= PXN || prox > 0.0)) { prox = prox * 4.0 > PX0 ? prox * 4.0 : PX0; if (prox > PROX_MAX) prox = PROX_MAX; }
            else if (alpha > 0.5) prox = prox > PX0 * 1e-3 ? prox / 4.0 : 0.0;
        }
        /* ---- update */
        for (int j = 0; j < 2; j++) w->nu_eq[j] += alpha * (w->nu_new[j] - w->nu_eq[j]);
        for (int k = 0; k <= N; k++) {
            if (k > 0) for (int j = 0; j < nx; j++) { w->X[k][j] += alpha * w->dX[k][j]; w->lam[k][j] += alpha * (w->lamn[k][j] - w->lam[k][j]); }
            if (k < N) for (int j = 0; j < nu; j++) w->U[k][j] += alpha * w->dU[k][j];
            w->s[k] += alpha * w->ds[k];
            for (int r = 0; r < w->nrow; r++) if (w->act[k][r]) { w->t[k][r] += alpha * w->dt_[k][r]; if (!z_done) w->z[k][r] += ad * w->dz[k][r]; }
        }
#undef NEWTON_SOLVE
#undef TRIAL_POINT
#undef ACCEPT_TEST
    }
    for (int k = 0; k <= N; k++) {
        for (int j = 0; j < nx; j++) Xo[k * nx + j] = w->X[k][j];
        if (k < N) for (int j = 0; j < nu; j++) Uo[k * nu + j] = w->U[k][j];
        so[k] = w->s[k];
    }
    if (iters_out) *iters_out = it;
    if (cost_out) *cost_out = cost_fn(w, w->X, w->U, w->s);
    if (err_out) *err_out = LAB[26] >= 1.0 ? mu : E0;   /* ([26]: experiments read the barrier parameter reached) */
    (void)nf;
    free(w); free(sv);
    return status;
}

/* batch driver (OpenMP over instances when compiled with -fopenmp).  obs stride = M*3 or (N+1)*M*3 */
int mmpc_oracle_solve_batch_guess(const oracle_cfg *cfg, int B, const double *x_init, const double *traj_ref, const double *u_ref,
                                  const double *u_last, const double *X0, const double *U0, const double *obs, double *Xo, double *Uo,
                                  double *so, int *status, int *iters, double *cost, double *err, int nthreads);
int mmpc_oracle_solve_batch(const oracle_cfg *cfg, int B, const double *x_init, const double *traj_ref, const double *u_ref,
                            const double *u_last, const double *X0, const double *obs, double *Xo, double *Uo, double *so,
                            int *status, int *iters, double *cost, double *err, int nthreads) {
    return mmpc_oracle_solve_batch_guess(cfg, B, x_init, traj_ref, u_ref, u_last, X0, 0, obs, Xo, Uo, so, status, iters, cost, err, nthreads);
}
/* U0 (or NULL): initial U separate from the U_last parameter - the opt-in warm start of include/mmpc.h (mmpc_set_warm_start) */
int mmpc_oracle_solve_batch_guess(const oracle_cfg *cfg, int B, const double *x_init, const double *traj_ref, const double *u_ref,
                                  const double *u_last, const double *X0, const double *U0, const double *obs, double *Xo, double *Uo,
                                  double *so, int *status, int *iters, double *cost, double *err, int nthreads) {
    int nx = cfg->kind == 0 ? 9 : 6, nu = cfg->kind == 0 ? 5 : 2, N = cfg->N;
    size_t so_ = (size_t)(cfg->obs_per_stage ? (N + 1) : 1) * cfg->M * 3;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int b = 0; b < B; b++) {
        status[b] = mmpc_oracle_solve(cfg, x_init + (size_t)b * nx, traj_ref + (size_t)b * (N + 1) * (cfg->pose_ref ? 4 : nx), u_ref + (size_t)b * N * nu,
                                      u_last + (size_t)b * N * nu, X0 ? X0 + (size_t)b * (N + 1) * nx : 0,
                                      U0 ? U0 + (size_t)b * N * nu : 0, obs + (size_t)b * so_,
                                      Xo + (size_t)b * (N + 1) * nx, Uo + (size_t)b * N * nu, so + (size_t)b * (N + 1),
                                      iters + b, cost + b, err + b);
    }
    return 0;
}

size_t mmpc_oracle_cfg_size(void) { return sizeof(oracle_cfg); }
