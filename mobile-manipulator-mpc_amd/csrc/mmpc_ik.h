// mmpc_ik.h - batched inverse kinematics of the planar 3-DoF arm: one lane per instance.
//
// Replaces ManipulatorPanda3DoF.inverse_transformation (robot_models/manipulator_3DoF.py:79-133), which hands
//     min_q (x_e(q) - x*)^2 + (z_e(q) - z*)^2   s.t.  -pi/2 <= q1 <= pi/2, -3pi/4 <= q2 <= 0, 0 <= q3 <= 3pi/2
// (cost :111, bounds :123) to IPOPT, started at q_initial_guess.  Here: projected Newton with Levenberg-Marquardt damping
// on the same objective (exact Hessian: the arm is three rotating segments, so d2 s_i / d theta_i^2 = -s_i).  The
// problem has 3 unknowns and 2 residuals: for a reachable target the minimisers form a curve and which point of it a
// solver returns depends on the solver (any of them is "the" answer of the reference, :83-86); for a target out of reach
// the minimiser is unique (arm stretched towards the target), which is what the reference's own known-answer value
// pins (utils/numerical_solve.py:5, manipulator_3DoF.py:219).
#pragma once
#include "mmpc_core.h"

#define MMPC_IK_MAXIT 200
#define MMPC_IK_OK 0
#define MMPC_IK_MAXITER 1

struct MmpcIkEval {
    double f, g[3], H[6];   // objective, gradient, packed lower Hessian
};

// segments s_i (rho, zeta) of the chain, angles theta = (q1, q1-q2, q1-q2-q3)  (manipulator_3DoF.py:30-70)
MMPC_DEV void mmpc_ik_eval(const double q[3], double tx, double tz, MmpcIkEval *e, bool derivs) {
    const double th[3] = {q[0], q[0] - q[1], q[0] - q[1] - q[2]};
    double sr[3], sz[3];
    { const double s = sin(th[0]), c = cos(th[0]); sr[0] = MMPC_A2 * s + MMPC_A3 * c; sz[0] = MMPC_A2 * c - MMPC_A3 * s; }
    { const double s = sin(th[1]), c = cos(th[1]); sr[1] = -MMPC_A3 * c + MMPC_A5 * s; sz[1] = MMPC_A3 * s + MMPC_A5 * c; }
    { const double s = sin(th[2]), c = cos(th[2]); sr[2] = MMPC_A6 * c - MMPC_A7 * s; sz[2] = -MMPC_A6 * s - MMPC_A7 * c; }
    const double rx = sr[0] + sr[1] + sr[2] - tx, rz = sz[0] + sz[1] + sz[2] - tz;
    e->f = rx * rx + rz * rz;
    if (!derivs) return;
    // d s_i / d theta_i = (zeta_i, -rho_i); d theta / d q: rows (1,0,0), (1,-1,0), (1,-1,-1)
    const double D[3][3] = {{1, 0, 0}, {1, -1, 0}, {1, -1, -1}};
    double Jx[3] = {0, 0, 0}, Jz[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++)
        for (int a = 0; a < 3; a++) { Jx[a] += sz[i] * D[i][a]; Jz[a] -= sr[i] * D[i][a]; }
    for (int a = 0; a < 3; a++) e->g[a] = 2.0 * (Jx[a] * rx + Jz[a] * rz);
    for (int a = 0; a < 3; a++)
        for (int b = 0; b <= a; b++) {
            double h = Jx[a] * Jx[b] + Jz[a] * Jz[b];
            for (int i = 0; i < 3; i++) h -= (rx * sr[i] + rz * sz[i]) * D[i][a] * D[i][b];
            e->H[a * (a + 1) / 2 + b] = 2.0 * h;
        }
}

// projected gradient (max norm): components that push against an active bound do not count
MMPC_DEV double mmpc_ik_pg(const double q[3], const double g[3], const double lo[3], const double hi[3], bool fr[3]) {
    double pg = 0.0;
    for (int a = 0; a < 3; a++) {
        fr[a] = !((q[a] <= lo[a] && g[a] > 0.0) || (q[a] >= hi[a] && g[a] < 0.0));
        if (fr[a]) pg = mmpc_max(pg, fabs(g[a]));
    }
    return pg;
}

// returns status (MMPC_IK_OK: projected gradient <= 1e-8); q_out = minimiser reached from q0 (clipped into the box first)
MMPC_DEV int mmpc_ik_solve(const double q0[3], double tx, double tz, double q_out[3], int *iters_out) {
    const double pi = 3.14159265358979323846;
    const double lo[3] = {-pi / 2, -3 * pi / 4, 0.0}, hi[3] = {pi / 2, 0.0, 3 * pi / 2};   // manipulator_3DoF.py:123
    double q[3];
    for (int a = 0; a < 3; a++) q[a] = mmpc_min(mmpc_max(q0[a], lo[a]), hi[a]);
    double lam = 1e-4;
    int it = 0;
    bool fr[3];
    MmpcIkEval e;
    mmpc_ik_eval(q, tx, tz, &e, true);
    for (it = 0; it < MMPC_IK_MAXIT; it++) {
        if (mmpc_ik_pg(q, e.g, lo, hi, fr) <= 1e-13) break;
        bool moved = false;
        double step = 0.0;
        for (int tries = 0; tries < 40 && !moved; tries++) {
            // (H_FF + lam I) p = -g_F by Cholesky; pinned variables get p = 0
            double A[6], p[3], L[6];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b <= a; b++)
                    A[a * (a + 1) / 2 + b] = (fr[a] && fr[b]) ? e.H[a * (a + 1) / 2 + b] + (a == b ? lam : 0.0) : (a == b ? 1.0 : 0.0);
            for (int a = 0; a < 3; a++) p[a] = fr[a] ? -e.g[a] : 0.0;
            bool pd = true;
            for (int j = 0; j < 3; j++) {
                double d = A[j * (j + 1) / 2 + j];
                for (int k = 0; k < j; k++) d -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
                if (!(d > 0.0)) { pd = false; break; }
                L[j * (j + 1) / 2 + j] = sqrt(d);
                for (int i = j + 1; i < 3; i++) {
                    double v = A[i * (i + 1) / 2 + j];
                    for (int k = 0; k < j; k++) v -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
                    L[i * (i + 1) / 2 + j] = v / L[j * (j + 1) / 2 + j];
                }
            }
            if (!pd) { lam = mmpc_max(10.0 * lam, 1e-8); continue; }
            for (int i = 0; i < 3; i++) { for (int k = 0; k < i; k++) p[i] -= L[i * (i + 1) / 2 + k] * p[k]; p[i] /= L[i * (i + 1) / 2 + i]; }
            for (int i = 2; i >= 0; i--) { for (int k = i + 1; k < 3; k++) p[i] -= L[k * (k + 1) / 2 + i] * p[k]; p[i] /= L[i * (i + 1) / 2 + i]; }
            double qn[3], pred = 0.0;
            step = 0.0;
            for (int a = 0; a < 3; a++) {
                qn[a] = mmpc_min(mmpc_max(q[a] + p[a], lo[a]), hi[a]);
                pred += e.g[a] * (qn[a] - q[a]);
                step = mmpc_max(step, fabs(qn[a] - q[a]));
            }
            MmpcIkEval en;
            mmpc_ik_eval(qn, tx, tz, &en, false);
            if (en.f <= e.f + 1e-4 * pred) {
                for (int a = 0; a < 3; a++) q[a] = qn[a];
                lam = mmpc_max(0.1 * lam, 1e-12);
                moved = true;
            } else lam = mmpc_max(10.0 * lam, 1e-8);
        }
        if (!moved) break;                       // no descent left at working precision
        mmpc_ik_eval(q, tx, tz, &e, true);
        if (step <= 1e-15) break;
    }
    for (int a = 0; a < 3; a++) q_out[a] = q[a];
    if (iters_out) *iters_out = it;
    return mmpc_ik_pg(q, e.g, lo, hi, fr) <= 1e-8 ? MMPC_IK_OK : MMPC_IK_MAXITER;
}
