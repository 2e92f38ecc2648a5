// mmpc_core.h - one-wavefront-per-problem interior-point MPC solver (gfx950).
//
// Replaces, for a batch of independent problem instances, what the reference does in
// MPCWholeBody.solve() / MPCBase.solve() (controllers/mpc_wholebody_qref.py:287-331,
// controllers/mpc_base.py:191-226): solve the multiple-shooting NLP that reset() builds
// (mpc_wholebody_qref.py:142-285, mpc_base.py:114-189) from the initial point
// X = tile(x_init) (or the previous X for the base kind), U = u_latest, s = 0.
// The reference delegates to CasADi -> IPOPT -> MUMPS; this is a direct primal-dual
// interior-point loop whose KKT system is factorised stage by stage (Riccati recursion).
//
// Execution model: ONE 64-lane wavefront (= one workgroup) owns one problem.  All solver
// state lives in that workgroup's LDS slab.  The code is a sequence of PHASES: inside a
// phase every lane works on its own slice (a stage, a matrix entry, a column) and reads
// only data produced by earlier phases; phases are separated by a workgroup barrier.
// No lane-private value crosses a phase boundary except wave-uniform scalars that every
// lane computes identically.  That discipline lets the very same source be compiled
//   (a) by hipcc for gfx950 (MMPC_EMU undefined): LANES_BEGIN/END = "this lane" + barrier
//   (b) by g++ with -DMMPC_EMU: LANES_BEGIN/END = a loop over the 64 lanes
// (b) exists only so that tests can run the kernel logic under ASAN/UBSAN on the CPU before
// it is launched on a GPU; it is never part of the product library.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef MMPC_EMU
#define MMPC_DEV inline
#define MMPC_HD inline
#define MMPC_CONST static const
#define LANES_BEGIN for (int lane = emu.first; lane != emu.end; lane += emu.step) {
#define LANES_BEGIN_NL LANES_BEGIN
#define LANES_END }
struct MmpcEmu { int first, end, step; };
#define MMPC_EMU_ARG , MmpcEmu emu
#else
#define MMPC_DEV __device__ __forceinline__
#define MMPC_HD __host__ __device__ inline
#define MMPC_CONST __device__ __constant__ static const
// The lane id is re-materialised through an opaque asm in every phase: otherwise LLVM hoists every
// lane-derived address out of the solver's loops and the hoisted values alone exceed the register file.
__device__ __forceinline__ int mmpc_lane_id() { int l = (int)threadIdx.x; asm volatile("" : "+v"(l)); return l; }
#define LANES_BEGIN { const int lane = mmpc_lane_id();
// a phase whose device code does not use the lane id (register-only blocks inside the stage loops: the opaque copy is one v_mov each)
#define LANES_BEGIN_NL {
// End of a phase.  A workgroup is ONE wavefront, and the LDS unit executes the LDS instructions of a wavefront in issue
// order: what one lane wrote in a phase is visible to every lane's reads of the next phase without waiting for the
// writes to retire.  So no s_barrier and no s_waitcnt here - only a fence that keeps the COMPILER from moving memory
// accesses across the phase boundary.  (-DMMPC_PHASE_SYNC restores the conservative __syncthreads() for A/B checks.)
#ifdef MMPC_PHASE_SYNC
#define LANES_END } __syncthreads();
#else
#define LANES_END } __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
#define MMPC_EMU_ARG
#endif

#define MMPC_WAVE 64
#define MMPC_FCAP 16
#define MMPC_MAX_LS 20
#include "mmpc_tile.h"
// Diagnostic build only (-DMMPC_STAMP_GEN): per-phase wave-cycle accounting of the generic kernel (tools/probe_stamps_generic.py)
#if defined(MMPC_STAMP_GEN) && !defined(MMPC_EMU)
__device__ unsigned long long mmpc_gstamp_acc[16];
#define MMPC_G0() unsigned long long g_prev_ = __builtin_readcyclecounter(), g_now_, g_acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, g_acc2_[6] = {0, 0, 0, 0, 0, 0};
#define MMPC_GS(i) { g_now_ = __builtin_readcyclecounter(); g_acc_[i] += g_now_ - g_prev_; g_prev_ = g_now_; }
// (slots 10..15: pieces of the Riccati pass - the terminal block R0, the two MFMA chains, the hand-over of the next stage's operands, the
//  elimination legs, the P store, gains + border columns; they are INSIDE slot 3's interval and subtract from it)
#define MMPC_G2() unsigned long long h_prev_ = __builtin_readcyclecounter(), h_now_;
#if defined(MMPC_STAMP_GEN_E)   // (slots 10..15 = pieces of the evaluation E1: cost + dynamics, box rows, circle rows, self rows, half-space rows, rows as written + tail)
#define MMPC_GS2(i)
#define MMPC_GSA(i)
#define MMPC_GSE(i) { h_now_ = __builtin_readcyclecounter(); g_acc2_[i] += h_now_ - h_prev_; h_prev_ = h_now_; }
#elif defined(MMPC_STAMP_GEN_A)   // (slots 10..15 = pieces of the assembly instead: cost + dynamics curvature, box rows, circle + self rows, half-space rows, rows as written, A2)
#define MMPC_GS2(i)
#define MMPC_GSE(i)
#define MMPC_GSA(i) { h_now_ = __builtin_readcyclecounter(); g_acc2_[i] += h_now_ - h_prev_; h_prev_ = h_now_; }
#else
#define MMPC_GSE(i)
#define MMPC_GS2(i) { h_now_ = __builtin_readcyclecounter(); g_acc2_[i] += h_now_ - h_prev_; h_prev_ = h_now_; }
#define MMPC_GSA(i)
#endif
#define MMPC_GEND() { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 10; i_++) atomicAdd(&mmpc_gstamp_acc[i_], g_acc_[i_]); for (int i_ = 0; i_ < 6; i_++) atomicAdd(&mmpc_gstamp_acc[10 + i_], g_acc2_[i_]); } }
#else
#define MMPC_G0()
#define MMPC_GS(i)
#define MMPC_G2()
#define MMPC_GS2(i)
#define MMPC_GSA(i)
#define MMPC_GSE(i)
#define MMPC_GEND()
#endif

// Per-launch constants (device memory, read through the scalar cache).
struct MmpcParams {
    int N, M, obs_per_stage, max_iter, use_xguess /* unused: the X guess is a launch argument (null = tile(x_init)) */, terminal_xy_eq;
    double dt, tol, mu_init, S;
    double Q2[81], P2[81];   // Q+Q^T, P+P^T (row-major, leading dimension NX)
    double RW2[25];          // (R+R^T)+(W+W^T), leading dimension NU
    double R2[25], W2[25];   // R+R^T, W+W^T
    double ulim[2][5], xlim[2][9], dulim[2][5];
    int L;                   // half-space ("manipulation") obstacles (whole-body kind), 0..8
    double hs[8][6];         // point (3), normal (3)   (demo_wholebody_qref.py:21-33)
    // opt-in (mmpc_set_warm_start): initial guess of U as [B][N][nu], separate from the U_last parameter of the cost and of
    // the rate bounds; null = the reference's protocol (U starts at U_last, mpc_wholebody_qref.py:303,310)
    const double *u_guess;
    // L >= 2 planes: also the L-1 extra rows per (stage >= 1, arm point) that obsAvoidConvex emits AS WRITTEN (quirk Q8,
    // mpc_wholebody_qref.py:77-89,156): -max(c_{k,i,0..j}, c_{k-1,i,j+1..L-1}) <= s_k  (generic kernel only)
    int as_written;
};

// robot_models/manipulator_3DoF.py:18-22, mobile_manipulator.py:14-15, base.py:15,
// mpc_wholebody_qref.py:43
#define MMPC_A2 0.316
#define MMPC_A3 0.0825
#define MMPC_A5 0.384
#define MMPC_A6 0.088
#define MMPC_A7 0.107
#define MMPC_BX (-0.007)
#define MMPC_BZ (0.606 + 0.333)
#define MMPC_BASE_R 0.4
#define MMPC_SELF_R 0.05
// weight of the augmentation rho/2 |E dx_N - e|^2 of the terminal-xy equality inside the stage-wise factorisation: it
// does not change the Newton step (constant on the constraint) but lets the exact Hessian be used whenever it is
// positive definite ON the constraint (without it the recursion falls back to Gauss-Newton -> linear convergence)
#define MMPC_RHO_EQ 1e4
// bound push of the initial point (IPOPT's bound_push): a variable with a finite simple bound starts at least this far
// inside it; the box rows are linear, so their slack then stays equal to the distance to the bound
#define MMPC_BOUND_PUSH 1e-2
// proximal term for crawling iterations: after two consecutive steps with alpha < MMPC_PROX_LO the (x,u) Hessian gets
// + prox I (MMPC_PROX0, x4 per further small step, at most MMPC_PROX_MAX), divided by 4 after a step with alpha > 0.5 (oracle/ipm_numpy.py)
#define MMPC_PROX0 100.0
// inertia correction (IPOPT: 1e-4, x100 the first time, x8 after, 1/3: the corrections this NLP needs are 1 ... 100)
#define MMPC_IC_D0 1.0
#define MMPC_IC_UP 4.0
#define MMPC_IC_DN (1.0 / 3.0)
// an iteration whose predecessor needed a correction above this starts from the corrected matrix at once (IPOPT always tries
// delta_w = 0 first: in a run of corrected iterations that attempt fails nearly every time - a wasted pass; 0.75 fewer passes per solve)
#define MMPC_IC_SKIP0 0.1
// second-order corrections per iteration (IPOPT: max_soc = 4), tried where theta(x_k) <= theta_min
#ifndef MMPC_SOC_MAX
#define MMPC_SOC_MAX 2
#endif
#define MMPC_PROX_LO 0.05
#define MMPC_PROX_MAX 1e4
// multiplier safeguard (IPOPT eq. 16): z_i is kept within [mu / (kappa t_i), kappa mu / t_i] at every evaluation
#define MMPC_KAPPA_SIGMA 1e10

// ---- structure of [A B] = d f / d(x,u) for the diff-drive base (+ integrator arm) -----------
// robot_models/base.py:19-26, manipulator_3DoF.py:190.  Every column has at most 4 non-zeros.
// Coefficient ids index the per-stage vector CV[k][11]:
//   0:0  1:1  2:dt  3:a32=-dt*u0*sin  4:a42=dt*u0*cos  5:a43=dt*dpsi  6:a34=-dt*dpsi
//   7:a35=-dt*dy  8:a45=dt*dx  9:dt*cos  10:dt*sin
#define MMPC_NCV 11
MMPC_CONST unsigned char kColRowWB[14][4] = {
    {0, 0, 0, 0}, {1, 0, 0, 0}, {2, 3, 4, 0}, {3, 0, 4, 0}, {4, 1, 3, 0}, {5, 2, 3, 4}, {6, 0, 0, 0},
    {7, 0, 0, 0}, {8, 0, 0, 0}, {3, 4, 0, 0}, {5, 0, 0, 0}, {6, 0, 0, 0}, {7, 0, 0, 0}, {8, 0, 0, 0}};
MMPC_CONST unsigned char kColCvWB[14][4] = {
    {1, 0, 0, 0}, {1, 0, 0, 0}, {1, 3, 4, 0}, {1, 2, 5, 0}, {1, 2, 6, 0}, {1, 2, 7, 8}, {1, 0, 0, 0},
    {1, 0, 0, 0}, {1, 0, 0, 0}, {9, 10, 0, 0}, {2, 0, 0, 0}, {2, 0, 0, 0}, {2, 0, 0, 0}, {2, 0, 0, 0}};
MMPC_CONST unsigned char kColRowB[8][4] = {
    {0, 0, 0, 0}, {1, 0, 0, 0}, {2, 3, 4, 0}, {3, 0, 4, 0}, {4, 1, 3, 0}, {5, 2, 3, 4}, {3, 4, 0, 0}, {5, 0, 0, 0}};
MMPC_CONST unsigned char kColCvB[8][4] = {
    {1, 0, 0, 0}, {1, 0, 0, 0}, {1, 3, 4, 0}, {1, 2, 5, 0}, {1, 2, 6, 0}, {1, 2, 7, 8}, {9, 10, 0, 0}, {2, 0, 0, 0}};
// rows of [A B] (forward roll-out): column index into [dx;du] and coefficient id, <= 5 terms
MMPC_CONST unsigned char kRowColWB[9][5] = {
    {0, 3, 0, 0, 0}, {1, 4, 0, 0, 0}, {2, 5, 0, 0, 0}, {3, 2, 4, 5, 9}, {4, 2, 3, 5, 9}, {5, 10, 0, 0, 0},
    {6, 11, 0, 0, 0}, {7, 12, 0, 0, 0}, {8, 13, 0, 0, 0}};
MMPC_CONST unsigned char kRowCvWB[9][5] = {
    {1, 2, 0, 0, 0}, {1, 2, 0, 0, 0}, {1, 2, 0, 0, 0}, {1, 3, 6, 7, 9}, {1, 4, 5, 8, 10}, {1, 2, 0, 0, 0},
    {1, 2, 0, 0, 0}, {1, 2, 0, 0, 0}, {1, 2, 0, 0, 0}};
MMPC_CONST unsigned char kRowColB[6][5] = {
    {0, 3, 0, 0, 0}, {1, 4, 0, 0, 0}, {2, 5, 0, 0, 0}, {3, 2, 4, 5, 6}, {4, 2, 3, 5, 6}, {5, 7, 0, 0, 0}};
MMPC_CONST unsigned char kRowCvB[6][5] = {
    {1, 2, 0, 0, 0}, {1, 2, 0, 0, 0}, {1, 2, 0, 0, 0}, {1, 3, 6, 7, 9}, {1, 4, 5, 8, 10}, {1, 2, 0, 0, 0}};
// packed lower-triangular index e -> (i,j), i>=j  (e = i(i+1)/2 + j), up to 9x9
MMPC_CONST unsigned char kTriI[45] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 6, 6,
                                      6, 6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 8};
MMPC_CONST unsigned char kTriJ[45] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5, 0, 1,
                                      2, 3, 4, 5, 6, 0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7, 8};
// self-collision check points p_i = alpha j2 + beta j3 (mpc_wholebody_qref.py:219)
MMPC_CONST double kSelfAB[4][2] = {{0.0, 0.0}, {0.5, 0.0}, {1.0, 0.0}, {0.5, 0.5}};
// state entries the forward kinematics depends on: x, y, psi, q1, q2, q3
MMPC_CONST unsigned char kY[6] = {0, 1, 2, 6, 7, 8};

template <int KIND>
struct MmpcDims {
    // KIND 0: whole-body, joint-space reference (controllers/mpc_wholebody_qref.py); 1: base only (controllers/mpc_base.py);
    // 2: whole-body, endpoint-pose reference (controllers/mpc_wholebody.py: 4-vector reference rows, no self-collision rows)
    static constexpr int NX = KIND != 1 ? 9 : 6;
    static constexpr int NU = KIND != 1 ? 5 : 2;
    static constexpr int NSELF = KIND == 0 ? 4 : 0;
    static constexpr int NREF = KIND == 2 ? 4 : NX;
    static constexpr int NV = NX + NU;
    static constexpr int NXX = NX * (NX + 1) / 2;
    static constexpr int NUU = NU * (NU + 1) / 2;
};

// per-launch constants kept in LDS (WTS block): the weights and the limits.  Read through the parameter block they are
// scalar loads that LLVM hoists out of every loop - several hundred live SGPRs, most of them spilled
#define MMPC_W_Q2 0      // [81]
#define MMPC_W_P2 81     // [81]
#define MMPC_W_RW2 162   // [25]
#define MMPC_W_R2 187    // [25]
#define MMPC_W_W2 212    // [25]
#define MMPC_W_XLIM 237  // [2][9]
#define MMPC_W_ULIM 255  // [2][5]
#define MMPC_W_DULIM 265 // [2][5]
#define MMPC_W_SIZE 276
#ifndef MMPC_GEN_FWD_UNROLL
#define MMPC_GEN_FWD_UNROLL 2   // (static shapes; 1 / 2 / 4: C4 batch 30.5 / 29.9 / 29.8 ms)
#endif
#ifndef MMPC_GEN_RIC_UNROLL
#define MMPC_GEN_RIC_UNROLL 2
#endif
#ifndef MMPC_GEN_TILE
#define MMPC_GEN_TILE 1   // the generic kernel's Riccati pass on v_mfma_f64_16x16x4_f64 tiles (as the specialised kernel's); 0: scalar pass through LDS
#endif
#define MMPC_NBC 3   // border columns: terminal multipliers nu0, nu1 and the slack s_{N-1} of the NLP as written (see A2)
// LDS slab layout (offsets in doubles).  Shared by host (size query) and device.
struct MmpcLayout {
    int X, U, S, LAM, XREF, UREF, ULAST, OBS, T, Z, HR, DTR, GC, HC, GSF, CV, CD, GX, GU, HXX, QX, HUXL, HUUL,
        HUX02, HUUD, QU, HSS, GSS, VX, VXN, KK, KF, DX, DU, DS, DLAM, PF, TT, PC, MF, MG, MH, MGX, MGU, RED, FILT,
        MISC, PNU, PNUS, KFV, GNU, FWV, NUEQ, GHS, WTS, GQ8, BQ8, VQ, RQ, HQX, QQX, HUXS, HUUS, RDX, SIGW, KU, PIV, total;
    int R, NR;
};

template <int KIND>
MMPC_HD constexpr MmpcLayout mmpc_layout(int N, int M, int obs_per_stage, int nhs = 0, int nq = 0) {
    typedef MmpcDims<KIND> D;
    MmpcLayout L{};
    int o = 0;
    const int NS = N + 1;
    // nhs = 6 half-space rows per stage when L > 0; nq = 6 (L-1) rows of the NLP as written (quirk Q8; none at stage 0)
    L.R = 2 * D::NU + 2 * D::NX + M + D::NSELF + nhs + nq;
    L.NR = M + D::NSELF + nhs + nq;
#define MMPC_CARVE(name, n) L.name = o; o += (n); o = (o + 1) & ~1;
    MMPC_CARVE(X, NS * D::NX) MMPC_CARVE(U, N * D::NU) MMPC_CARVE(S, NS) MMPC_CARVE(LAM, NS * D::NX)
    MMPC_CARVE(XREF, NS * D::NX) MMPC_CARVE(UREF, N * D::NU) MMPC_CARVE(ULAST, N * D::NU)
    MMPC_CARVE(OBS, (obs_per_stage ? NS : 1) * M * 3)
    MMPC_CARVE(T, NS * L.R) MMPC_CARVE(Z, NS * L.R) MMPC_CARVE(HR, NS * L.NR) MMPC_CARVE(DTR, NS * L.NR)
    MMPC_CARVE(GC, NS * M * 2) MMPC_CARVE(HC, NS * M * 3) MMPC_CARVE(GSF, NS * D::NSELF * 6)
    MMPC_CARVE(CV, NS * MMPC_NCV) MMPC_CARVE(CD, NS * D::NX) MMPC_CARVE(GX, NS * D::NX) MMPC_CARVE(GU, NS * D::NU)
    MMPC_CARVE(HXX, NS * D::NXX) MMPC_CARVE(QX, NS * D::NX) MMPC_CARVE(HUXL, D::NU * D::NX) MMPC_CARVE(HUUL, D::NUU)
    MMPC_CARVE(HUX02, NS) MMPC_CARVE(HUUD, NS * D::NU) MMPC_CARVE(QU, NS * D::NU) MMPC_CARVE(HSS, NS)
    MMPC_CARVE(GSS, NS) MMPC_CARVE(VX, NS * 6) MMPC_CARVE(VXN, 6 + D::NX + D::NU)
    MMPC_CARVE(KK, N * D::NU * D::NX) MMPC_CARVE(KF, N * D::NU)
    MMPC_CARVE(DX, NS * D::NX) MMPC_CARVE(DU, NS * D::NU) MMPC_CARVE(DS, NS) MMPC_CARVE(DLAM, NS * D::NX)
    MMPC_CARVE(PF, D::NX * D::NX) MMPC_CARVE(TT, D::NX * D::NV) MMPC_CARVE(PC, D::NX) MMPC_CARVE(MF, D::NXX)
    MMPC_CARVE(MG, D::NU * D::NX) MMPC_CARVE(MH, D::NUU) MMPC_CARVE(MGX, D::NX) MMPC_CARVE(MGU, D::NU)
    MMPC_CARVE(RED, 8 * MMPC_WAVE) MMPC_CARVE(FILT, 2 * MMPC_FCAP) MMPC_CARVE(MISC, 8)
    // border variables of the stage-wise system (MMPC_NBC columns): the two multipliers of the terminal xy equality
    // (interface_wholebody_qref.py:166-167) and the slack s_{N-1} when it reaches back to x_{N-2} (NLP as written): sensitivities
    MMPC_CARVE(PNU, D::NX * MMPC_NBC) MMPC_CARVE(PNUS, NS * D::NX * MMPC_NBC) MMPC_CARVE(KFV, N * D::NU * MMPC_NBC) MMPC_CARVE(GNU, D::NV * MMPC_NBC)
    MMPC_CARVE(FWV, 2 * (1 + MMPC_NBC) * D::NX) MMPC_CARVE(NUEQ, 4) MMPC_CARVE(GHS, NS * nhs * 6) MMPC_CARVE(WTS, MMPC_W_SIZE) MMPC_CARVE(SIGW, 16)
    MMPC_CARVE(KU, MMPC_GEN_TILE ? N * (D::NU * (D::NU - 1) / 2) : 0) MMPC_CARVE(PIV, MMPC_GEN_TILE ? N * D::NU : 0)
    // as-written rows: gradient / branch per row; per stage: coupling of s_k to x_{k-1}, what stage k's rows add to stage k-1
    // (residual, Hessian y-block, gradient), dense blocks of a slack eliminated one stage earlier, x-stationarity residual
    MMPC_CARVE(GQ8, NS * nq * 6) MMPC_CARVE(BQ8, NS * nq) MMPC_CARVE(VQ, nq ? NS * 6 : 0) MMPC_CARVE(RQ, nq ? (NS + 1) * 6 : 0)
    MMPC_CARVE(HQX, nq ? (NS + 1) * 21 : 0) MMPC_CARVE(QQX, nq ? (NS + 1) * 6 : 0) MMPC_CARVE(HUXS, nq ? NS * D::NU * D::NX : 0)
    MMPC_CARVE(HUUS, nq ? NS * D::NUU : 0) MMPC_CARVE(RDX, nq ? NS * D::NX : 0)
#undef MMPC_CARVE
    L.total = o;
    return L;
}

struct MmpcIO {
    const double *x_init, *traj_ref, *u_ref, *u_last, *x_guess, *obs;  // this problem's slices
    const double *u_guess;                                             // (or null)
    double *X, *U, *s, *cost, *err;
    int *status, *iters;
    // iteration budget / continuation (specialised kernels): after `budget` iterations of this launch (0: no budget) an
    // unconverged instance writes its primal-dual state to `state` and ends with status MMPC_STATUS_SUSPENDED; a launch
    // with resume != 0 starts from that state and runs exactly the iterations the uninterrupted solve would have run next
    double *state;
    int budget, resume;
    double *gscr;      // this instance's gain block in global memory (specialised kernels of long horizons, MmpcGainBlock), else unused
    double *soc;       // this instance's scratch of the second-order correction in global memory (mmpc_soc_doubles; null: no corrections)
};
// doubles of an instance's second-order-correction scratch (both kernels; NR = rows per stage that are not box rows): the
// uncorrected direction, and for the specialised kernels - which move to a trial point in place - the constraint residuals of the
// trial point and the running c_soc
MMPC_HD constexpr int mmpc_soc_doubles(int N, int NX, int NU, int NR) { return (N + 1) * (4 * NX + NU + 1 + 2 * NR) + 8; }

MMPC_DEV double mmpc_min(double a, double b) { return a < b ? a : b; }
MMPC_DEV double mmpc_bound_push(double v, double lo, double hi) {
    double lo2 = lo + MMPC_BOUND_PUSH, hi2 = hi - MMPC_BOUND_PUSH;
    if (lo2 > hi2) lo2 = hi2 = 0.5 * (lo + hi);
    return v < lo2 ? lo2 : (v > hi2 ? hi2 : v);
}
MMPC_DEV double mmpc_max(double a, double b) { return a > b ? a : b; }
// the same for running error maxima: a NaN on EITHER side stays (mmpc_max hands only a NaN in b through), so that a
// non-finite residual anywhere reaches the status 2 test of the main loop
MMPC_DEV double mmpc_max_err(double a, double b) { return (a > b || a != a) ? a : b; }
// Reductions of the generic kernel over the MMPC_WAVE per-lane partials a phase left in LDS: one LDS read per lane and a butterfly
// through DPP / permlane swaps (mmpc_tile.h) instead of every lane walking through all 64 words (~10 k cycles per reduction site,
// five to eight sites per iteration).  With the operations of this file (a NaN partial reaches the result of MAXERR).
#ifdef MMPC_EMU
#define MMPC_GRED_SUM(ARR) mmpc_emu_red_arr((ARR), 0)
#define MMPC_GRED_MAX(ARR) mmpc_emu_red_arr((ARR), 1)
#define MMPC_GRED_MIN(ARR) mmpc_emu_red_arr((ARR), 2)
#define MMPC_GRED_MAXERR(ARR) mmpc_emu_red_arr((ARR), 3)
#else
#if MMPC_RED_DPP
MMPC_DEV double mmpc_gop_add(double a, double b) { return a + b; }
MMPC_WAVE_RED(mmpc_gwave_sum, mmpc_gop_add)
MMPC_WAVE_RED(mmpc_gwave_max, mmpc_max)
MMPC_WAVE_RED(mmpc_gwave_min, mmpc_min)
MMPC_WAVE_RED(mmpc_gwave_maxerr, mmpc_max_err)
#else
#error "the generic kernel's reductions need MMPC_RED_DPP"
#endif
#define MMPC_GRED_SUM(ARR) mmpc_gwave_sum((ARR)[mmpc_lane_id()])
#define MMPC_GRED_MAX(ARR) mmpc_gwave_max((ARR)[mmpc_lane_id()])
#define MMPC_GRED_MIN(ARR) mmpc_gwave_min((ARR)[mmpc_lane_id()])
#define MMPC_GRED_MAXERR(ARR) mmpc_gwave_maxerr((ARR)[mmpc_lane_id()])
#endif
MMPC_DEV double mmpc_z_safeguard(double z, double t, double mu) {
    const double p = z * t;
    if (p > MMPC_KAPPA_SIGMA * mu) return MMPC_KAPPA_SIGMA * mu / t;
    if (p < mu / MMPC_KAPPA_SIGMA) return mu / (MMPC_KAPPA_SIGMA * t);
    return z;
}
// (ap: the fraction-to-boundary step of the iteration - the crawl the term is for is the one that rule causes; a step the line
//  search cut is the business of the inertia correction; alpha: the accepted step)
MMPC_DEV void mmpc_prox_update(double ap, double alpha, double &prox, int &nsmall) {
    nsmall = ap < MMPC_PROX_LO ? nsmall + 1 : 0;
    if (ap < MMPC_PROX_LO && (nsmall >= 2 || prox > 0.0)) prox = mmpc_min(MMPC_PROX_MAX, mmpc_max(MMPC_PROX0, 4.0 * prox));
    else if (alpha > 0.5) prox = prox > MMPC_PROX0 * 1e-3 ? 0.25 * prox : 0.0;
}
MMPC_DEV bool mmpc_finite(double v) { return fabs(v) < 1.0e300; }
// a simple bound counts as absent from 1e19 on, as in IPOPT (nlp_lower_bound_inf / nlp_upper_bound_inf = -+1e19; the reference
// leaves them at their defaults): a "large number for infinity" does not become a row - and the slack of such a row would be the one
// factor that can overflow the product form of the barrier's log sum (MmpcLogAcc)
MMPC_DEV bool mmpc_bound_active(double b) { return fabs(b) < 1.0e19; }

// ---- light-weight sincos (both kernels; the generic one used the library routine until round 3) ------------
// sin/cos: Cody-Waite reduction by pi/2 (three-part constant, exact for |x| < ~1e6 rad: heading and
// joint angles stay far below that) + the fdlibm kernel polynomials on [-pi/4, pi/4] (< 1 ulp).
// The library sincos carries a Payne-Hanek path that costs ~700 instructions per call site.
MMPC_DEV void mmpc_sincos(double x, double *sn, double *cs) {
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = fma(-n, 1.57079632673412561417e+00, x);
    r = fma(-n, 6.07710050630396597660e-11, r);
    r = fma(-n, 2.02226624879595063154e-21, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                      2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03),
                      -1.66666666666666324348e-01);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                      -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03),
                      4.16666666666666019037e-02);
    const double s0 = fma(r * z, ps, r);
    const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)n & 3;
    const double sa = (q & 1) ? c0 : s0, ca = (q & 1) ? s0 : c0;
    *sn = (q & 2) ? -sa : sa;
    *cs = ((q + 1) & 2) ? -ca : ca;
}
#ifndef MMPC_LIBM_SINCOS
#define MMPC_SINCOS(x, s, c) mmpc_sincos((x), (s), (c))
#else
#define MMPC_SINCOS(x, s, c) sincos((x), (s), (c))
#endif
// planar arm segments, manipulator_3DoF.py:29-73 collapsed with A=q1-q2, B=q1-q2-q3
MMPC_DEV void mmpc_arm_segments(double q1, double q2, double q3, double dr[3], double dz[3]) {
    double s1, c1, sA, cA, sB, cB;
    MMPC_SINCOS(q1, &s1, &c1);
    MMPC_SINCOS(q1 - q2, &sA, &cA);
    MMPC_SINCOS(q1 - q2 - q3, &sB, &cB);
    dr[0] = MMPC_A2 * s1 + MMPC_A3 * c1;
    dz[0] = MMPC_A2 * c1 - MMPC_A3 * s1;
    dr[1] = -MMPC_A3 * cA + MMPC_A5 * sA;
    dz[1] = MMPC_A3 * sA + MMPC_A5 * cA;
    dr[2] = MMPC_A6 * cB - MMPC_A7 * sB;
    dz[2] = -MMPC_A6 * sB - MMPC_A7 * cB;
}

// n = sqrt(m) and inv = -1 / (2 n).  Device: v_rsq_f64 + one cubic step and one correction of the root (<= 1-2 ulp) - the
// IEEE sqrt and division sequences are ~75 instructions per row, this is 11
MMPC_DEV void mmpc_sqrt_pair(double m, double *n, double *inv) {
#ifdef MMPC_EMU
    *n = sqrt(m); *inv = -1.0 / (2 * *n);
#else
    const double y0 = __builtin_amdgcn_rsq(m), e = fma(-(m * y0), y0, 1.0);
    const double y = fma(y0 * e, fma(0.375, e, 0.5), y0);   // cubic step: 1.4e-16 (tools/rcp_probe.hip)
    const double r = m * y;
    *n = fma(fma(-r, r, m), 0.5 * y, r);
    *inv = -0.5 * y;
#endif
}

// self-collision row i (mpc_wholebody_qref.py:219-222): h = 0.05 - ||alpha j2 + beta j3 - e||
// (world points).  g6 (may be null) = dh/d(x,y,psi,q1,q2,q3).
MMPC_DEV double mmpc_self_row(int i, double px, double py, double c, double s, const double dr[3], const double dz[3],
                              double *g6) {
    // p_i = alpha j2 + beta j3: (0,0), (1/2,0), (1,0), (1/2,1/2)   (mpc_wholebody_qref.py:219)
    const double al = i == 0 ? 0.0 : (i == 2 ? 1.0 : 0.5), be = i == 3 ? 0.5 : 0.0, kap = al + be - 1.0;
    const double cm0 = kap, cm1 = be - 1.0, cm2 = -1.0;
    const double R = kap * MMPC_BX + cm0 * dr[0] + cm1 * dr[1] + cm2 * dr[2];
    const double Z = kap * MMPC_BZ + cm0 * dz[0] + cm1 * dz[1] + cm2 * dz[2];
    const double C = px * c + py * s, Dv = -px * s + py * c;
    const double mm = kap * kap * (px * px + py * py) + 2 * kap * R * C + R * R + Z * Z;
    double n, inv;
    mmpc_sqrt_pair(mm, &n, &inv);
    if (g6) {
        // segment m rotates with angle a_m.q, a_1=(1,0,0), a_2=(1,-1,0), a_3=(1,-1,-1)
        const double z0 = cm0 * dz[0], z1 = cm1 * dz[1], z2 = cm2 * dz[2];
        const double r0 = cm0 * dr[0], r1 = cm1 * dr[1], r2 = cm2 * dr[2];
        const double Ri0 = z0 + z1 + z2, Ri1 = -z1 - z2, Ri2 = -z2;
        const double Zi0 = -(r0 + r1 + r2), Zi1 = r1 + r2, Zi2 = r2;
        const double f = 2 * (kap * C + R);
        g6[0] = (2 * kap * kap * px + 2 * kap * R * c) * inv;
        g6[1] = (2 * kap * kap * py + 2 * kap * R * s) * inv;
        g6[2] = (2 * kap * R * Dv) * inv;
        g6[3] = (f * Ri0 + 2 * Z * Zi0) * inv;
        g6[4] = (f * Ri1 + 2 * Z * Zi1) * inv;
        g6[5] = (f * Ri2 + 2 * Z * Zi2) * inv;
    }
    return MMPC_SELF_R - n;
}

// half-space row for arm sample point i of [j2/2, j2, (j2+j3)/2, j3, (j3+e)/2, e]
// (mpc_wholebody_qref.py:57-89, 216-217):  h = -max_j n_j.((pi_j - 0.03 n_j) - P_i(x)).
// One row per (stage, point): the intended formulation; the reference's L>=2 code path emits L rows that read
// stale / free `constr` entries (quirk Q8) - not reproduced.  g6 = dh/d(x,y,psi,q1,q2,q3) or null.
MMPC_DEV double mmpc_hs_row_range(const MmpcParams &P, int i, int j0, int j1, double px, double py, double c, double s,
                                  const double dr[3], const double dz[3], double *g6, double *h10);
MMPC_DEV double mmpc_hs_row(const MmpcParams &P, int i, double px, double py, double c, double s, const double dr[3],
                            const double dz[3], double *g6, double *h10 = nullptr) {
    return mmpc_hs_row_range(P, i, 0, P.L, px, py, c, s, dr, dz, g6, h10);
}
// the same over the planes j0 <= j < j1 only: -max_j n_j.((pi_j - 0.03 n_j) - P_i(x))
MMPC_DEV double mmpc_hs_row_range(const MmpcParams &P, int i, int j0, int j1, double px, double py, double c, double s,
                                  const double dr[3], const double dz[3], double *g6, double *h10) {
    const double al = (i == 0 || i == 2) ? 0.5 : (i == 1 ? 1.0 : 0.0);
    const double be = (i == 2 || i == 4) ? 0.5 : (i == 3 ? 1.0 : 0.0);
    const double ga = i == 4 ? 0.5 : (i == 5 ? 1.0 : 0.0);
    const double sig = al + be + ga, cm0 = sig, cm1 = be + ga, cm2 = ga;
    const double R = sig * MMPC_BX + cm0 * dr[0] + cm1 * dr[1] + cm2 * dr[2];
    const double Z = sig * MMPC_BZ + cm0 * dz[0] + cm1 * dz[1] + cm2 * dz[2];
    const double Pw0 = sig * px + R * c, Pw1 = sig * py + R * s, Pw2 = Z;
    double best = 0.0;
    int jb = -1;
    for (int j = j0; j < j1; j++) {
        const double *h = P.hs[j];
        const double v = h[3] * ((h[0] - 0.03 * h[3]) - Pw0) + h[4] * ((h[1] - 0.03 * h[4]) - Pw1) + h[5] * ((h[2] - 0.03 * h[5]) - Pw2);
        if (jb < 0 || v > best) { best = v; jb = j; }
    }
    if (g6) {
        const double n0 = P.hs[jb][3], n1 = P.hs[jb][4], n2 = P.hs[jb][5];
        const double z0 = cm0 * dz[0], z1 = cm1 * dz[1], z2 = cm2 * dz[2];
        const double r0 = cm0 * dr[0], r1 = cm1 * dr[1], r2 = cm2 * dr[2];
        const double Rm0 = z0 + z1 + z2, Rm1 = -z1 - z2, Rm2 = -z2;
        const double Zm0 = -(r0 + r1 + r2), Zm1 = r1 + r2, Zm2 = r2;
        g6[0] = sig * n0; g6[1] = sig * n1; g6[2] = n0 * (-R * s) + n1 * (R * c);
        g6[3] = (n0 * c + n1 * s) * Rm0 + n2 * Zm0;
        g6[4] = (n0 * c + n1 * s) * Rm1 + n2 * Zm1;
        g6[5] = (n0 * c + n1 * s) * Rm2 + n2 * Zm2;
        if (h10) {
            // second derivatives of n.P_i(x) over (psi, q1, q2, q3), packed lower triangle: d2 seg/d theta^2 = -seg, so the
            // entries of R_qq are +-Zm and those of Z_qq are -+Rm (oracle/nlp.py: halfspace_row, order 2)
            const double nc = n0 * c + n1 * s, nt = -n0 * s + n1 * c;
            h10[0] = -nc * R;
            h10[1] = nt * Rm0; h10[2] = nc * Zm0 - n2 * Rm0;
            h10[3] = nt * Rm1; h10[4] = nc * Zm1 - n2 * Rm1; h10[5] = -nc * Zm1 + n2 * Rm1;
            h10[6] = nt * Rm2; h10[7] = nc * Zm2 - n2 * Rm2; h10[8] = -nc * Zm2 + n2 * Rm2; h10[9] = -nc * Zm2 + n2 * Rm2;
        }
    }
    return -best;
}

MMPC_DEV double mmpc_angle_diff(double a, double b) {  // mpc_base.py:56-94
    const double PI = 3.14159265358979323846;
    a = fmod(a + PI, 2 * PI) - PI;
    b = fmod(b + PI, 2 * PI) - PI;
    const double d = a - b;
    if (a * b >= 0) return d;
    if (a > b) return d <= PI ? d : d - 2 * PI;
    return d > -PI ? d : d + 2 * PI;
}

template <int KIND>
struct MmpcTab;
template <>
struct MmpcTab<0> {
    MMPC_DEV static int crow(int j, int q) { return kColRowWB[j][q]; }
    MMPC_DEV static int ccv(int j, int q) { return kColCvWB[j][q]; }
    MMPC_DEV static int rcol(int i, int q) { return kRowColWB[i][q]; }
    MMPC_DEV static int rcv(int i, int q) { return kRowCvWB[i][q]; }
};
template <>
struct MmpcTab<2> : MmpcTab<0> {};
template <>
struct MmpcTab<1> {
    MMPC_DEV static int crow(int j, int q) { return kColRowB[j][q]; }
    MMPC_DEV static int ccv(int j, int q) { return kColCvB[j][q]; }
    MMPC_DEV static int rcol(int i, int q) { return kRowColB[i][q]; }
    MMPC_DEV static int rcv(int i, int q) { return kRowCvB[i][q]; }
};

// State term of a stage: returns q = e^T W2 e with W2 = W + W^T (the cost term is q/2), W = Q (k < N) or P; optional
// gradient (NX) and packed lower Hessian (Gauss-Newton part, plus the remaining curvature when `exact`).
//   KIND 0 / 1: e = x - ref (angleDiff on psi for the base kind, mpc_base.py:148), Hessian = W2;
//   KIND 2: e = endpoint pose (x + R cos psi, y + R sin psi, Z, psi) - ref (controllers/mpc_wholebody.py:79-80,104-106 with
//           mobile_manipulator.py:36-53), W2 is 4x4 (leading dimension 4); Hessian = J^T W2 J + sum_c (W2 e)_c d2E_c.
template <int KIND>
MMPC_DEV double mmpc_state_cost(const double *WTS, bool terminal, const double *xk, const double *ref, double *grad,
                                double *hxx, bool exact) {
    typedef MmpcDims<KIND> D;
    constexpr int NX = D::NX;
    const double *W2 = WTS + (terminal ? MMPC_W_P2 : MMPC_W_Q2);
    if constexpr (KIND != 2) {
        double e[NX], q = 0.0;
        for (int j = 0; j < NX; j++) e[j] = xk[j] - ref[j];
        if (KIND == 1) e[2] = mmpc_angle_diff(xk[2], ref[2]);
        for (int i = 0; i < NX; i++) {
            double v = 0.0;
            for (int j = 0; j < NX; j++) v += W2[i * NX + j] * e[j];
            if (grad) grad[i] = v;
            q += e[i] * v;
        }
        if (hxx) for (int e2 = 0; e2 < D::NXX; e2++) hxx[e2] = W2[kTriI[e2] * NX + kTriJ[e2]];
        (void)exact;
        return q;
    } else {
    double sn, cs, dr[3], dz[3];
    MMPC_SINCOS(xk[2], &sn, &cs);
    mmpc_arm_segments(xk[NX - 3], xk[NX - 2], xk[NX - 1], dr, dz);
    const double R = MMPC_BX + dr[0] + dr[1] + dr[2], Z = MMPC_BZ + dz[0] + dz[1] + dz[2];
    const double e[4] = {xk[0] + R * cs - ref[0], xk[1] + R * sn - ref[1], Z - ref[2], xk[2] - ref[3]};
    double v[4], q = 0.0;
    for (int i = 0; i < 4; i++) { v[i] = 0.0; for (int j = 0; j < 4; j++) v[i] += W2[i * 4 + j] * e[j]; q += e[i] * v[i]; }
    if (!grad && !hxx) return q;
    const double Rm[3] = {dz[0] + dz[1] + dz[2], -(dz[1] + dz[2]), -dz[2]}, Zm[3] = {-(dr[0] + dr[1] + dr[2]), dr[1] + dr[2], dr[2]};
    double J[4][6];   // over y = (x, y, psi, q1, q2, q3)
    for (int i = 0; i < 4; i++) for (int a = 0; a < 6; a++) J[i][a] = 0.0;
    J[0][0] = 1.0; J[0][2] = -R * sn; J[1][1] = 1.0; J[1][2] = R * cs; J[3][2] = 1.0;
    for (int a = 0; a < 3; a++) { J[0][3 + a] = Rm[a] * cs; J[1][3 + a] = Rm[a] * sn; J[2][3 + a] = Zm[a]; }
    if (grad) {
        for (int j = 0; j < NX; j++) grad[j] = 0.0;
        for (int a = 0; a < 6; a++) { double g = 0.0; for (int i = 0; i < 4; i++) g += J[i][a] * v[i]; grad[kY[a]] = g; }
    }
    if (hxx) {
        for (int e2 = 0; e2 < D::NXX; e2++) hxx[e2] = 0.0;
        for (int a = 0; a < 6; a++) for (int b = 0; b <= a; b++) {
            double h = 0.0;
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) h += J[i][a] * W2[i * 4 + j] * J[j][b];
            hxx[kY[a] * (kY[a] + 1) / 2 + kY[b]] = h;
        }
        if (exact) {
            // same closed form as the half-space rows with n = (v0, v1, v2): entries of R_qq are +-Zm, of Z_qq -+Rm
            const double nc = v[0] * cs + v[1] * sn, nt = -v[0] * sn + v[1] * cs;
            const double h10[10] = {-nc * R,
                                    nt * Rm[0], nc * Zm[0] - v[2] * Rm[0],
                                    nt * Rm[1], nc * Zm[1] - v[2] * Rm[1], -nc * Zm[1] + v[2] * Rm[1],
                                    nt * Rm[2], nc * Zm[2] - v[2] * Rm[2], -nc * Zm[2] + v[2] * Rm[2], -nc * Zm[2] + v[2] * Rm[2]};
            for (int a = 0; a < 4; a++) for (int b = 0; b <= a; b++)
                hxx[kY[2 + a] * (kY[2 + a] + 1) / 2 + kY[2 + b]] += h10[a * (a + 1) / 2 + b];
        }
    }
    return q;
    }
}

// ------------------------------------------------------------------------------------------
// The solver.  `lds` points at this problem's slab of mmpc_layout<KIND>(N,M,..).total doubles.
// ------------------------------------------------------------------------------------------
// NC, MC, OPSC, LC: horizon, circle obstacles, per-stage obstacle table and number of half-space planes as compile-time
// constants (0 / -1: taken from P at run time).  With run-time sizes the 54 LDS offsets of the layout are live scalars for
// the whole solve - they spill to vector lanes, and those to scratch; instantiated for a shape they are constants.
// NC, MC, OPSC, LC, AWC: horizon, circle obstacles, obs_per_stage, half-space planes and the as-written flag when they are
// constants of the instantiation (the demo's shapes: every LDS offset is then an immediate), 0 / -1: read from the parameter block
#if MMPC_GEN_TILE
// lane state of the Riccati pass on MFMA tiles (lane = 16 g + j; accumulator register r <-> row g + 4 r, column j of the
// 16x16 tile over (x, 1, u)); the same fields as the specialised kernel's MmpcLaneState (mmpc_fast.h)
template <int KIND>
struct MmpcGenRic {
    typedef MmpcDims<KIND> D;
    static constexpr int NKB = (D::NX + 1 + 3) / 4;           // K-blocks (4 rows each) that cover the (x, 1) rows
    static constexpr int NPU = D::NU * (D::NU - 1) / 2;       // couplings between the inputs of a stage
    static constexpr bool PAIRS = ((D::NX + 1) & 1) == 0;     // first input row even: rows (2q, 2q+1) share a register
    static constexpr int NLEG = PAIRS ? (D::NU + 1) / 2 : D::NU;
    static_assert(D::NV + 1 <= 16, "stage matrix over (x, 1, u) must fit one 16x16 tile");
    unsigned ab_o[NKB];                  // [A B c; 0 0 1] operand rows 4r+g, column j: LDS offset | stage stride << 16
    int p_o[4], p_s[4];                  // where register r of [P_k | p_k] is stored (offset at stage 0, stage stride; a dump slot otherwise)
    unsigned fw_st, fw_in;               // forward roll-out, row `lane` of [A B]: up to four state terms (column 4 bits | coefficient id 4 bits each);
                                         // input of the row + 1 | coefficient id << 8 | (this lane stores the input step) << 16
    unsigned fwb_st, fwb_in;             // the same for row `lane % NX` (roll-out of the border columns: lane = c NX + i)
    unsigned bdA, bdB;                   // border columns: the <= 4 (row, coefficient id) pairs (4 + 4 bits each) of the dynamics column this lane
                                         // handles in the input part (lane = c NU + a) and in the state part (lane = c NX + i) of their recursion
    unsigned st_o[4][4];                 // stage-matrix entry of register r = sum of up to four LDS words: offset | stage stride << 16 | (only at the last stage) << 31
    int kl_b[NLEG], kl_s[NLEG];          // where the lane stores its entry of the normalised pivot row(s) of leg l
    MmpcAcc rP, rT, rM;                  // cost-to-go [P p; p^T .], its product with the dynamics, stage matrix
    double rAB[NKB], opa, opb;           // MFMA operands
    double nab[NKB], nhm[4];             // next stage's dynamics rows and stage-matrix entries
};
#endif

template <int KIND, int NC = 0, int MC = -1, int OPSC = -1, int LC = -1, int AWC = -1>
MMPC_DEV void mmpc_solve_one(const MmpcParams &P, const MmpcIO io, double *lds MMPC_EMU_ARG) {
    typedef MmpcDims<KIND> D;
    typedef MmpcTab<KIND> TB;
    constexpr int NX = D::NX, NU = D::NU, NSELF = D::NSELF, NV = D::NV, NXX = D::NXX, NUU = D::NUU;
    const int N = NC ? NC : P.N, M = MC >= 0 ? MC : P.M, NS = N + 1;
    const int OPS = OPSC >= 0 ? OPSC : P.obs_per_stage, PL = LC >= 0 ? LC : P.L;
    const int NHS = (KIND == 0 && PL > 0) ? 6 : 0;
    const int NQ = (KIND == 0 && (AWC >= 0 ? AWC : P.as_written) && PL >= 2) ? 6 * (PL - 1) : 0;   // rows of the NLP as written (quirk Q8), stages >= 1
    constexpr int NREF = D::NREF;
    constexpr int HRMAX = KIND == 0 ? 16 + 4 + 6 + 42 : 16 + 4 + 6;
    const MmpcLayout L = mmpc_layout<KIND>(N, M, OPS, NHS, NQ);
    const int R = L.R, NR = L.NR;
    double *X = lds + L.X, *U = lds + L.U, *S = lds + L.S, *LAM = lds + L.LAM, *XREF = lds + L.XREF,
           *UREF = lds + L.UREF, *ULAST = lds + L.ULAST, *OBS = lds + L.OBS, *T = lds + L.T, *Z = lds + L.Z,
           *HR = lds + L.HR, *DTR = lds + L.DTR, *GC = lds + L.GC, *HC = lds + L.HC, *GSF = lds + L.GSF,
           *CV = lds + L.CV, *CD = lds + L.CD, *GX = lds + L.GX, *GU = lds + L.GU, *HXX = lds + L.HXX,
           *QX = lds + L.QX, *HUXL = lds + L.HUXL, *HUUL = lds + L.HUUL, *HUX02 = lds + L.HUX02,
           *HUUD = lds + L.HUUD, *QU = lds + L.QU, *HSS = lds + L.HSS, *GSS = lds + L.GSS, *VX = lds + L.VX,
           *VXN = lds + L.VXN, *KK = lds + L.KK, *KF = lds + L.KF, *DX = lds + L.DX, *DU = lds + L.DU,
           *DS = lds + L.DS, *DLAM = lds + L.DLAM, *PF = lds + L.PF, *TT = lds + L.TT, *PC = lds + L.PC,
           *MF = lds + L.MF, *MG = lds + L.MG, *MH = lds + L.MH, *MGX = lds + L.MGX, *MGU = lds + L.MGU,
           *RED = lds + L.RED, *FILT = lds + L.FILT, *MISC = lds + L.MISC, *PNU = lds + L.PNU, *PNUS = lds + L.PNUS,
           *KFV = lds + L.KFV, *GNU = lds + L.GNU, *FWV = lds + L.FWV, *NUEQ = lds + L.NUEQ, *GHS = lds + L.GHS, *SIGW = lds + L.SIGW, *KU = lds + L.KU, *PIV = lds + L.PIV;
    double *const WTS = lds + L.WTS;
    double *const GQ8 = lds + L.GQ8, *const BQ8 = lds + L.BQ8, *const VQ = lds + L.VQ, *const RQ = lds + L.RQ, *const HQX = lds + L.HQX,
           *const QQX = lds + L.QQX, *const HUXS = lds + L.HUXS, *const HUUS = lds + L.HUUS, *const RDX = lds + L.RDX;
    const bool teq = P.terminal_xy_eq != 0;
    const int SL_UHI = NU, SL_XLO = 2 * NU, SL_XHI = 2 * NU + NX, SL_C = 2 * NU + 2 * NX, SL_S = SL_C + M, SL_H = SL_S + NSELF, SL_Q = SL_H + NHS;
    const double dt = P.dt, Sw = P.S, tol = P.tol;

    // bound of a box slot r at stage k; returns false when the row does not exist
    auto box_bound = [&](int k, int r, double &b) -> bool {
        if (r < SL_XLO) {
            if (k >= N) return false;
            const int j = r < NU ? r : r - NU;
            const double ul = ULAST[k * NU + j];
            if (r < NU) b = mmpc_max(WTS[MMPC_W_ULIM + j], ul + WTS[MMPC_W_DULIM + j]);
            else b = mmpc_min(WTS[MMPC_W_ULIM + 5 + j], ul + WTS[MMPC_W_DULIM + 5 + j]);
        } else {
            if (k < 1) return false;
            const int q = r - SL_XLO;
            b = q < NX ? WTS[MMPC_W_XLIM + q] : WTS[MMPC_W_XLIM + 9 + q - NX];
        }
        return mmpc_bound_active(b);
    };
    auto obs_ptr = [&](int k, int m) -> const double * {
        return OBS + ((OPS ? k * M : 0) + m) * 3;
    };
    auto slack_idx = [&](int k) -> int { return k < N - 1 ? k : N - 1; };  // :265 quirk (Q1)
    // row e = i (L-1) + j of the NLP as written at stage k >= 1: min(-max_{j' <= j} c_{k,i,j'}, -max_{j' > j} c_{k-1,i,j'}) (without
    // the - s_k); br = 1 when the previous stage's planes attain it - value, gradient and curvature then belong to x_{k-1}
    auto q8_row = [&](int e, const double *xk, double cs, double sn, const double *dr, const double *dz, const double *xp, double csp,
                      double snp, const double *drp, const double *dzp, int &br, double *g6, double *h10) -> double {
        const int i = e / (PL - 1), j = e % (PL - 1);
        const double vc = mmpc_hs_row_range(P, i, 0, j + 1, xk[0], xk[1], cs, sn, dr, dz, nullptr, nullptr);
        const double vp = mmpc_hs_row_range(P, i, j + 1, PL, xp[0], xp[1], csp, snp, drp, dzp, nullptr, nullptr);
        br = vp < vc ? 1 : 0;
        if (g6) {
            if (br) mmpc_hs_row_range(P, i, j + 1, PL, xp[0], xp[1], csp, snp, drp, dzp, g6, h10);
            else mmpc_hs_row_range(P, i, 0, j + 1, xk[0], xk[1], cs, sn, dr, dz, g6, h10);
        }
        return br ? vp : vc;
    };

    // ---------------------------------------------------------------- per-launch constants -> LDS
    LANES_BEGIN
    for (int i = lane; i < MMPC_W_SIZE; i += MMPC_WAVE) {
        double v;
        if (i < MMPC_W_P2) v = P.Q2[i];
        else if (i < MMPC_W_RW2) v = P.P2[i - MMPC_W_P2];
        else if (i < MMPC_W_R2) v = P.RW2[i - MMPC_W_RW2];
        else if (i < MMPC_W_W2) v = P.R2[i - MMPC_W_R2];
        else if (i < MMPC_W_XLIM) v = P.W2[i - MMPC_W_W2];
        else if (i < MMPC_W_ULIM) v = P.xlim[(i - MMPC_W_XLIM) / 9][(i - MMPC_W_XLIM) % 9];
        else if (i < MMPC_W_DULIM) v = P.ulim[(i - MMPC_W_ULIM) / 5][(i - MMPC_W_ULIM) % 5];
        else v = P.dulim[(i - MMPC_W_DULIM) / 5][(i - MMPC_W_DULIM) % 5];
        WTS[i] = v;
    }
    LANES_END
    // ---------------------------------------------------------------- load + initial point
    LANES_BEGIN
    for (int i = lane; i < NS * NX; i += MMPC_WAVE) {
        const int j = i % NX;
        double x0 = io.x_init[j];
        if (KIND != 1) { const double lo = WTS[MMPC_W_XLIM + j], hi = WTS[MMPC_W_XLIM + 9 + j]; x0 = x0 > hi ? hi : (x0 < lo ? lo : x0); }  // :290-291 (a NaN stays a NaN)
        if (i < NS * NREF) XREF[i] = io.traj_ref[i];
        X[i] = (io.x_guess && i >= NX) ? io.x_guess[i] : x0;  // :302 / mpc_base.py:200
        LAM[i] = 0.0;
    }
    for (int i = lane; i < N * NU; i += MMPC_WAVE) {
        UREF[i] = io.u_ref[i];
        ULAST[i] = io.u_last[i];
        U[i] = io.u_guess ? io.u_guess[i] : io.u_last[i];  // :303,:310
    }
    for (int i = lane; i < NS; i += MMPC_WAVE) S[i] = 0.0;  // :304
    if (lane < 4) NUEQ[lane] = 0.0;
    if (lane < 16) SIGW[lane] = 0.0;
    for (int i = lane; i < (OPS ? NS : 1) * M * 3; i += MMPC_WAVE) OBS[i] = io.obs[i];
    LANES_END
    // bound push of the initial point (needs U_last of the previous phase for the merged input box)
    LANES_BEGIN
    for (int i = lane; i < N * NU; i += MMPC_WAVE) {
        const int k = i / NU, j = i % NU;
        double lo, hi;
        if (!box_bound(k, j, lo)) lo = -INFINITY;
        if (!box_bound(k, NU + j, hi)) hi = INFINITY;
        U[i] = mmpc_bound_push(U[i], lo, hi);
    }
    for (int i = lane + NX; i < NS * NX; i += MMPC_WAVE) {
        const int k = i / NX, j = i % NX;
        double lo, hi;
        if (!box_bound(k, SL_XLO + j, lo)) lo = -INFINITY;
        if (!box_bound(k, SL_XLO + NX + j, hi)) hi = INFINITY;
        X[i] = mmpc_bound_push(X[i], lo, hi);
    }
    LANES_END

    // value of the non-box rows of stage k at a point (used for slack init and line search)
    // returns via hr[NR]
    auto nl_rows = [&](int k, const double *xk, double sk, double sks, double *hr) {
        for (int m = 0; m < M; m++) {
            const double *o = obs_ptr(k, m);
            const double dx = xk[0] - o[0], dy = xk[1] - o[1];
            hr[m] = (o[2] + MMPC_BASE_R) - sqrt(dx * dx + dy * dy) - sk;  // mpc_wholebody_qref.py:53
        }
        if (NSELF) {
            double dr[3], dz[3], sn, cs;
            MMPC_SINCOS(xk[2], &sn, &cs);
            mmpc_arm_segments(xk[NX - 3], xk[NX - 2], xk[NX - 1], dr, dz);
            for (int i = 0; i < NSELF; i++) hr[M + i] = mmpc_self_row(i, xk[0], xk[1], cs, sn, dr, dz, nullptr) - sks;
            for (int i = 0; i < NHS; i++) hr[M + NSELF + i] = mmpc_hs_row(P, i, xk[0], xk[1], cs, sn, dr, dz, nullptr) - sk;
            if (NQ && k >= 1) {
                const double *xp = X + (k - 1) * NX;
                double drp[3], dzp[3], snp, csp;
                MMPC_SINCOS(xp[2], &snp, &csp);
                mmpc_arm_segments(xp[NX - 3], xp[NX - 2], xp[NX - 1], drp, dzp);
                for (int e = 0; e < NQ; e++) { int br; hr[M + NSELF + NHS + e] = q8_row(e, xk, cs, sn, dr, dz, xp, csp, snp, drp, dzp, br, nullptr, nullptr) - sk; }
            }
        }
    };

    double mu = P.mu_init;
    // ---------------------------------------------------------------- slack / multiplier init
    LANES_BEGIN
    for (int k = lane; k < NS; k += MMPC_WAVE) {
        double hr[HRMAX];
        nl_rows(k, X + k * NX, S[k], S[slack_idx(k)], hr);
        for (int r = 0; r < R; r++) {
            double h = 0.0, b;
            bool act = true;
            if (r >= SL_Q && k == 0) act = false;   // (at k = 0 the stale entries are free variables: no row)
            else if (r < SL_C) {
                act = box_bound(k, r, b);
                if (act) {
                    if (r < NU) h = b - U[k * NU + r];
                    else if (r < SL_XLO) h = U[k * NU + r - NU] - b;
                    else if (r < SL_XHI) h = b - X[k * NX + r - SL_XLO];
                    else h = X[k * NX + r - SL_XHI] - b;
                }
            } else h = hr[r - SL_C];
            const double t0 = act ? mmpc_max(-h, 1e-2) : 1.0;
            T[k * R + r] = t0;
            Z[k * R + r] = act ? mu / t0 : 0.0;
        }
    }
    LANES_END

    int status = 1, it = 0, nfilt = 0, filt_init = 0, nrows_act = 0, nsmall = 0;
    double prox = 0.0, delta_last = 0.0, delta_prev = 0.0;   // (delta_prev: the correction of the previous iteration's matrix, 0 when it needed none)
#if MMPC_GEN_TILE
    // Riccati recursion on MFMA tiles over (x, 1, u), as in mmpc_fast.h (see there for the lane <-> entry map): lane l = 16 g + j
    // holds rows g + 4 r of column j in accumulator register r
    typedef MmpcGenRic<KIND> GR;
    constexpr int NKB = GR::NKB, NLEG = GR::NLEG, NPU = GR::NPU;
    constexpr bool PAIRS = GR::PAIRS;
#ifdef MMPC_EMU
    static thread_local GR ls_all[MMPC_WAVE];
#else
    GR ls_one;
#endif
    // tile index -> index in the (x, u | gradient) numbering: state t, gradient NV, input NX + a, 99 = outside
    auto tvar = [&](int t) -> int { return t < NX ? t : (t == NX ? NV : (t <= NV ? t - 1 : 99)); };
    LANES_BEGIN
    {
        auto &ls = MMPC_LS;
        const int g = lane >> 4, j = lane & 15, jl = tvar(j);
        for (int r = 0; r < NKB; r++) {
            // operand entry of [A B c; 0 0 1] (rows over (x, 1), columns over (x, 1, u)), row m = 4 r + g: a coefficient of CV[k]
            // (ids 0, 1, 2 are the constants 0, 1, dt), the defect c_k[m] in the column of the 1, the 1 itself, or the constant 0
            const int m = 4 * r + g;
            unsigned off = (unsigned)L.CV, stride = 0;
            if (m < NX && jl < NV) {
                int id = 0;
                for (int q = 0; q < 4; q++) if (TB::crow(jl, q) == m && TB::ccv(jl, q) != 0) id = TB::ccv(jl, q);
                off = (unsigned)(L.CV + id); stride = id >= 3 ? MMPC_NCV : 0;
            } else if (m < NX && jl == NV) { off = (unsigned)(L.CD + m); stride = NX; }
            else if (m == NX && jl == NV) off = (unsigned)(L.CV + 1);
            ls.ab_o[r] = off | (stride << 16);
        }
        for (int r = 0; r < 4; r++) {
            // where register r of [P_k | p_k] goes: lower triangle of P_k -> HXX[k], p_k -> QX[k], everything else a dump slot
            const int i = tvar(g + 4 * r);
            int off = L.RED + lane, stride = 0;
            if (i < NX && jl < NX && i >= jl) { off = L.HXX + i * (i + 1) / 2 + jl; stride = NXX; }
            else if (i < NX && jl == NV) { off = L.QX + i; stride = NX; }
            ls.p_o[r] = off; ls.p_s[r] = stride;
        }
        for (int r = 0; r < 4; r++) {
            // entry (ti, tj) of a stage's matrix [Hxx Hxu q_x; Hux Huu q_u; q^T 0] in tile numbering (what R2 of the scalar pass adds
            // to [A B]^T P [A B]), as the sum - in this order - of up to four words: packed Hxx; gradient in row / column NX;
            // Hux = dense block of a slack eliminated at the last stage (quirk Q1) + the same of the as-written rows (any stage) + the
            // (0,2) entry of the dynamics curvature; Huu = R2 + W2 + diagonal barrier terms + the two dense blocks.  An absent term
            // reads the constant 0 of the coefficient vector (CV[0]).
            int i = tvar(g + 4 * r), jj = jl;
            if (i < jj) { const int t_ = i; i = jj; jj = t_; }
            auto pk = [](int off, int stride, int lastonly) -> unsigned { return (unsigned)off | ((unsigned)stride << 16) | ((unsigned)lastonly << 31); };
            unsigned t[4];
            for (int q = 0; q < 4; q++) t[q] = pk(L.CV, 0, 0);
            if (i == 99) { }
            else if (i == NV) { if (jj < NX) t[0] = pk(L.QX + jj, NX, 0); else if (jj < NV) t[0] = pk(L.QU + jj - NX, NU, 0); }
            else if (i < NX) t[0] = pk(L.HXX + i * (i + 1) / 2 + jj, NXX, 0);
            else {
                const int a_ = i - NX;
                if (jj < NX) {
                    t[0] = pk(L.HUXL + a_ * NX + jj, 0, 1);
                    if (NQ) t[1] = pk(L.HUXS + a_ * NX + jj, NU * NX, 0);
                    if (a_ == 0 && jj == 2) t[2] = pk(L.HUX02, 1, 0);
                } else {
                    const int b_ = jj - NX, e2 = a_ * (a_ + 1) / 2 + b_;
                    t[0] = pk(L.WTS + MMPC_W_RW2 + a_ * NU + b_, 0, 0);
                    if (a_ == b_) t[1] = pk(L.HUUD + a_, NU, 0);
                    t[2] = pk(L.HUUL + e2, 0, 1);
                    if (NQ) t[3] = pk(L.HUUS + e2, NUU, 0);
                }
            }
            for (int q = 0; q < 4; q++) ls.st_o[r][q] = t[q];
        }
        {
            const int colA = NX + (NU > 0 ? lane % (NU > 0 ? NU : 1) : 0), colB = lane % NX;
            unsigned wa = 0, wb = 0;
            for (int q = 0; q < 4; q++) {
                if (lane < MMPC_NBC * NU) wa |= ((unsigned)TB::crow(colA, q) | ((unsigned)TB::ccv(colA, q) << 4)) << (8 * q);
                if (lane < MMPC_NBC * NX) wb |= ((unsigned)TB::crow(colB, q) | ((unsigned)TB::ccv(colB, q) << 4)) << (8 * q);
            }
            ls.bdA = wa; ls.bdB = wb;
        }
        {
            unsigned st = 0, fi = 0;
            if (lane < NX) {
                int ns = 0, na = -1, ida = 0;
                for (int q = 0; q < 5; q++) {
                    const int c = TB::rcol(lane, q), id = TB::rcv(lane, q);
                    if (id == 0) continue;
                    if (c < NX) { st |= ((unsigned)c | ((unsigned)id << 4)) << (8 * ns); ns++; }
                    else { na = c - NX; ida = id; }
                }
                bool writer = na >= 0;
                for (int l2 = 0; l2 < lane; l2++) for (int q = 0; q < 5; q++) if (TB::rcv(l2, q) != 0 && TB::rcol(l2, q) == NX + na) writer = false;
                fi = (unsigned)(na + 1) | ((unsigned)ida << 8) | ((writer ? 1u : 0u) << 16);
            }
            ls.fw_st = st; ls.fw_in = fi;
            unsigned stb = 0, fib = 0;
            {
                const int ib = lane % NX;
                int ns = 0, na = -1, ida = 0;
                for (int q = 0; q < 5; q++) {
                    const int c = TB::rcol(ib, q), id = TB::rcv(ib, q);
                    if (id == 0) continue;
                    if (c < NX) { stb |= ((unsigned)c | ((unsigned)id << 4)) << (8 * ns); ns++; }
                    else { na = c - NX; ida = id; }
                }
                fib = (unsigned)(na + 1) | ((unsigned)ida << 8);
            }
            ls.fwb_st = stb; ls.fwb_in = fib;
        }
        for (int l = 0; l < NLEG; l++) {
            // entry of the normalised pivot row(s) of leg l this lane holds: gain row (column < NX), feed-forward (column NX),
            // coupling to a later input, or nothing (dump slot)
            const int a0 = PAIRS ? 2 * l : l;
            int off = L.RED + lane, stride = 0;
            for (int a = a0; a < NU && a <= a0 + (PAIRS ? 1 : 0); a++) {
                const int ta = NX + 1 + a;
                if (g == (ta & 3)) {
                    if (j < NX) { off = L.KK + a * NX + j; stride = NU * NX; }
                    else if (j == NX) { off = L.KF + a; stride = NU; }
                    else if (j > ta && j <= NV) { const int b2 = j - NX - 1; off = L.KU + a * (2 * NU - a - 1) / 2 + (b2 - a - 1); stride = NPU; }
                }
            }
            ls.kl_b[l] = off; ls.kl_s[l] = stride;
        }
    }
    LANES_END
#endif
    double E0 = 0.0, th_max = 0.0, th_min = 0.0;
    // number of active rows (uniform): box rows that exist + all non-box rows
    for (int r = 0; r < SL_C; r++) {
        // u rows depend on k only through u_last (finite => same activity for all k unless dulim finite and
        // ulim infinite, still finite); count exactly:
        for (int k = 0; k < NS; k++) { double b; if (box_bound(k, r, b)) nrows_act++; }
    }
    nrows_act += NS * NR - NQ;

    MMPC_G0()
    for (it = 0; it <= P.max_iter; it++) {
        MMPC_GS(0)
        // ============================================================ E1: evaluation + KKT partials
        MMPC_G2()
        LANES_BEGIN
        double e_d = 0.0, e_p = 0.0, tzmax = 0.0, tzmin = 1e300, zsum = 0.0, zeq = 0.0;
        // the merit function at this point (what the line search compares its trials with) falls out of the same evaluation:
        // objective terms, l1 infeasibility, sum of log t (the barrier parameter may still change before the line search)
        double m_f = 0.0, m_th = 0.0, m_lg = 0.0;
        for (int k = lane; k < NS; k += MMPC_WAVE) {
            const double *xk = X + k * NX;
            double mant = 1.0; int ex = 0;   // product of the mantissas of the stage's slacks + sum of their exponents (one log per stage)
            auto acc = [&](double tv) { int e2; mant *= frexp(tv, &e2); ex += e2; if (mant < 1e-200) { mant = frexp(mant, &e2); ex += e2; } };
            double sn, cs;
            MMPC_SINCOS(xk[2], &sn, &cs);
            double rdx[NX], rdu[NU > 0 ? NU : 1];
            // cost gradient (mpc_wholebody_qref.py:192-201,240-242; mpc_base.py:146-153)
            {
                double g[NX];
                m_f += Sw * S[k] * S[k] + 0.5 * mmpc_state_cost<KIND>(WTS, k == N, xk, XREF + k * NREF, g, nullptr, false);
                for (int i = 0; i < NX; i++) { GX[k * NX + i] = g[i]; rdx[i] = g[i] + (k >= 1 ? LAM[k * NX + i] : 0.0); }
            }
            double *cv = CV + k * MMPC_NCV;
            if (k < N) {
                const double *uk = U + k * NU;
                cv[0] = 0.0; cv[1] = 1.0; cv[2] = dt;
                cv[3] = -dt * uk[0] * sn; cv[4] = dt * uk[0] * cs; cv[5] = dt * xk[5]; cv[6] = -dt * xk[5];
                cv[7] = -dt * xk[4]; cv[8] = dt * xk[3]; cv[9] = dt * cs; cv[10] = dt * sn;
                // defect c_k = f(x_k,u_k) - x_{k+1}   (base.py:19-26, manipulator_3DoF.py:190)
                double xn[NX];
                xn[0] = xk[0] + dt * xk[3]; xn[1] = xk[1] + dt * xk[4]; xn[2] = xk[2] + dt * xk[5];
                xn[3] = xk[3] + dt * (uk[0] * cs - xk[4] * xk[5]);
                xn[4] = xk[4] + dt * (uk[0] * sn + xk[3] * xk[5]);
                xn[5] = xk[5] + dt * uk[1];
                if (KIND != 1) { xn[6] = xk[6] + dt * uk[2]; xn[7] = xk[7] + dt * uk[3]; xn[8] = xk[8] + dt * uk[4]; }
                for (int j = 0; j < NX; j++) {
                    const double c = xn[j] - X[(k + 1) * NX + j];
                    CD[k * NX + j] = c;
                    e_p = mmpc_max_err(e_p, fabs(c));
                    m_th += fabs(c);
                    zeq += fabs(LAM[(k + 1) * NX + j]);
                }
                double qin = 0.0;
                for (int a = 0; a < NU; a++) {
                    double v = 0.0, vr = 0.0, vw = 0.0;
                    for (int b = 0; b < NU; b++) {
                        const double tr = WTS[MMPC_W_R2 + a * NU + b] * (uk[b] - UREF[k * NU + b]), tw = WTS[MMPC_W_W2 + a * NU + b] * (uk[b] - ULAST[k * NU + b]);
                        v += tr + tw; vr += tr; vw += tw;
                    }
                    GU[k * NU + a] = v;
                    rdu[a] = v;
                    qin += (uk[a] - UREF[k * NU + a]) * vr + (uk[a] - ULAST[k * NU + a]) * vw;
                }
                m_f += 0.5 * qin;
                // - A^T lam_{k+1}, - B^T lam_{k+1}
                const double *ln = LAM + (k + 1) * NX;
                for (int j = 0; j < NV; j++) {
                    double v = 0.0;
                    for (int q = 0; q < 4; q++) v += cv[TB::ccv(j, q)] * ln[TB::crow(j, q)];
                    if (j < NX) rdx[j] -= v; else rdu[j - NX] -= v;
                }
            }
            MMPC_GSE(0)
            // box rows, in the order of the slots (input lower / upper, state lower / upper); loops with compile-time bounds: the
            // residual entries a row adds to are then registers (indexed by a run-time row number they were a select chain per row)
            auto box_row = [&](int r, double h, double z) {
                const double t = T[k * R + r];
                e_p = mmpc_max_err(e_p, fabs(h + t)); m_th += fabs(h + t); acc(t);
                tzmax = mmpc_max(tzmax, t * z); tzmin = mmpc_min(tzmin, t * z); zsum += z;
            };
            if (k < N) {
                for (int a = 0; a < NU; a++) { double b; if (box_bound(k, a, b)) { const double z = Z[k * R + a]; rdu[a] -= z; box_row(a, b - U[k * NU + a], z); } }
                for (int a = 0; a < NU; a++) { double b; if (box_bound(k, NU + a, b)) { const double z = Z[k * R + NU + a]; rdu[a] += z; box_row(NU + a, U[k * NU + a] - b, z); } }
            }
            for (int j = 0; j < NX; j++) { double b; if (box_bound(k, SL_XLO + j, b)) { const double z = Z[k * R + SL_XLO + j]; rdx[j] -= z; box_row(SL_XLO + j, b - xk[j], z); } }
            for (int j = 0; j < NX; j++) { double b; if (box_bound(k, SL_XHI + j, b)) { const double z = Z[k * R + SL_XHI + j]; rdx[j] += z; box_row(SL_XHI + j, xk[j] - b, z); } }
            MMPC_GSE(1)
            // circle rows: value, gradient, Hessian (mpc_wholebody_qref.py:49-54)
            double rds = 2 * Sw * S[k], selfz = 0.0;
            for (int m = 0; m < M; m++) {
                const double *o = obs_ptr(k, m);
                const double dx = xk[0] - o[0], dy = xk[1] - o[1], d = sqrt(dx * dx + dy * dy), id = 1.0 / d;
                const double nxv = dx * id, nyv = dy * id;
                const double h = (o[2] + MMPC_BASE_R) - d - S[k];
                HR[k * NR + m] = h;
                GC[(k * M + m) * 2 + 0] = -nxv; GC[(k * M + m) * 2 + 1] = -nyv;
                HC[(k * M + m) * 3 + 0] = -(1 - nxv * nxv) * id; HC[(k * M + m) * 3 + 1] = nxv * nyv * id;
                HC[(k * M + m) * 3 + 2] = -(1 - nyv * nyv) * id;
                const double t = T[k * R + SL_C + m], z = Z[k * R + SL_C + m];
                rdx[0] -= nxv * z; rdx[1] -= nyv * z; rds -= z;
                e_p = mmpc_max_err(e_p, fabs(h + t)); m_th += fabs(h + t); acc(t);
                tzmax = mmpc_max(tzmax, t * z); tzmin = mmpc_min(tzmin, t * z); zsum += z;
            }
            MMPC_GSE(2)
            if (NSELF) {
                double dr[3], dz[3];
                mmpc_arm_segments(xk[NX - 3], xk[NX - 2], xk[NX - 1], dr, dz);
                const double sks = S[slack_idx(k)];
                for (int i = 0; i < NSELF; i++) {
                    double g6[6];
                    const double h = mmpc_self_row(i, xk[0], xk[1], cs, sn, dr, dz, g6) - sks;
                    HR[k * NR + M + i] = h;
                    const double t = T[k * R + SL_S + i], z = Z[k * R + SL_S + i];
                    for (int a = 0; a < 6; a++) { GSF[(k * NSELF + i) * 6 + a] = g6[a]; rdx[kY[a]] += g6[a] * z; }
                    selfz += z;
                    e_p = mmpc_max_err(e_p, fabs(h + t)); m_th += fabs(h + t); acc(t);
                    tzmax = mmpc_max(tzmax, t * z); tzmin = mmpc_min(tzmin, t * z); zsum += z;
                }
                MMPC_GSE(3)
                for (int i = 0; i < NHS; i++) {   // half-space rows, bound to s_k (s_N at the end, :268)
                    double g6[6];
                    const double h = mmpc_hs_row(P, i, xk[0], xk[1], cs, sn, dr, dz, g6) - S[k];
                    HR[k * NR + M + NSELF + i] = h;
                    const double t = T[k * R + SL_H + i], z = Z[k * R + SL_H + i];
                    for (int a = 0; a < 6; a++) { GHS[(k * 6 + i) * 6 + a] = g6[a]; rdx[kY[a]] += g6[a] * z; }
                    rds -= z;
                    e_p = mmpc_max_err(e_p, fabs(h + t)); m_th += fabs(h + t); acc(t);
                    tzmax = mmpc_max(tzmax, t * z); tzmin = mmpc_min(tzmin, t * z); zsum += z;
                }
                MMPC_GSE(4)
                if (NQ) {   // rows of the NLP as written (quirk Q8): planes 0..j at x_k, planes j+1.. at x_{k-1}
                    double rq[6] = {0, 0, 0, 0, 0, 0};
                    if (k >= 1) {
                        const double *xp = X + (k - 1) * NX;
                        double drp[3], dzp[3], snp, csp;
                        MMPC_SINCOS(xp[2], &snp, &csp);
                        mmpc_arm_segments(xp[NX - 3], xp[NX - 2], xp[NX - 1], drp, dzp);
                        for (int e = 0; e < NQ; e++) {
                            double g6[6];
                            int br;
                            const double h = q8_row(e, xk, cs, sn, dr, dz, xp, csp, snp, drp, dzp, br, g6, nullptr) - S[k];
                            HR[k * NR + M + NSELF + NHS + e] = h;
                            BQ8[k * NQ + e] = (double)br;
                            const double t = T[k * R + SL_Q + e], z = Z[k * R + SL_Q + e];
                            for (int a = 0; a < 6; a++) {
                                GQ8[(k * NQ + e) * 6 + a] = g6[a];
                                if (br) rq[a] += g6[a] * z; else rdx[kY[a]] += g6[a] * z;
                            }
                            rds -= z;
                            e_p = mmpc_max_err(e_p, fabs(h + t)); m_th += fabs(h + t); acc(t);
                            tzmax = mmpc_max(tzmax, t * z); tzmin = mmpc_min(tzmin, t * z); zsum += z;
                        }
                    }
                    for (int a = 0; a < 6; a++) RQ[k * 6 + a] = rq[a];   // what this stage's rows add to the stationarity of x_{k-1}
                    if (k == N) for (int a = 0; a < 6; a++) RQ[(N + 1) * 6 + a] = 0.0;
                }
            }
            if (k < N) rds -= selfz; else MISC[1] = selfz;
            DS[k] = rds;  // stage-local part of the s-stationarity residual (finished in E1b)
            if (teq && k == N) {   // X[N,:2] == X_ref[N,:2] with multiplier nu
                for (int j = 0; j < 2; j++) {
                    rdx[j] += NUEQ[j];
                    e_p = mmpc_max_err(e_p, fabs(xk[j] - XREF[N * NX + j]));
                    m_th += fabs(xk[j] - XREF[N * NX + j]);
                    zeq += fabs(NUEQ[j]);
                }
            }
            if (NQ) { for (int i = 0; i < NX; i++) RDX[k * NX + i] = rdx[i]; }   // finished in E1b (the next stage's rows add to it)
            else if (k >= 1) for (int i = 0; i < NX; i++) e_d = mmpc_max_err(e_d, fabs(rdx[i]));
            if (k < N) for (int a = 0; a < NU; a++) e_d = mmpc_max_err(e_d, fabs(rdu[a]));
            m_lg += log(mant) + (double)ex * 0.69314718055994530942;
            MMPC_GSE(5)
        }
        RED[6 * MMPC_WAVE + lane] = m_f; RED[7 * MMPC_WAVE + lane] = m_th; PF[lane] = m_lg;   // (PF, with TT behind it: >= 64 doubles, not in use before the backward pass)
        RED[0 * MMPC_WAVE + lane] = e_d; RED[1 * MMPC_WAVE + lane] = e_p; RED[2 * MMPC_WAVE + lane] = tzmax;
        RED[3 * MMPC_WAVE + lane] = tzmin; RED[4 * MMPC_WAVE + lane] = zsum;   // (zsum: the row multipliers alone)
        HXX[lane] = zeq;   // |equality multipliers|: the stage Hessians are not in use before the assembly
        if (lane == 0) MISC[0] = 0.0;
        if (NSELF == 0 && lane == 0) MISC[1] = 0.0;
        LANES_END
        // ---- E1b: finish the s residual (terminal self rows are bound to s_{N-1})
        LANES_BEGIN
        double e_s = 0.0;
        for (int k = lane; k < NS; k += MMPC_WAVE) e_s = mmpc_max_err(e_s, fabs(DS[k] - (k == N - 1 ? MISC[1] : 0.0)));
        if (NQ)
            for (int k = lane; k < NS; k += MMPC_WAVE) {
                if (k < 1) continue;
                double r9[NX];
                for (int i = 0; i < NX; i++) r9[i] = RDX[k * NX + i];
                for (int a = 0; a < 6; a++) r9[kY[a]] += RQ[(k + 1) * 6 + a];
                for (int i = 0; i < NX; i++) e_s = mmpc_max_err(e_s, fabs(r9[i]));
            }
        RED[5 * MMPC_WAVE + lane] = e_s;
        LANES_END
        double err_d = 0.0, err_p = 0.0, tzmax = 0.0, tzmin = 1e300, zsum = 0.0;
        double m_F0 = 0.0, m_TH0 = 0.0, m_LG0 = 0.0;
        m_F0 = MMPC_GRED_SUM(RED + 6 * MMPC_WAVE); m_TH0 = MMPC_GRED_SUM(RED + 7 * MMPC_WAVE); m_LG0 = MMPC_GRED_SUM(PF);
        err_d = mmpc_max_err(MMPC_GRED_MAXERR(RED + 0 * MMPC_WAVE), MMPC_GRED_MAXERR(RED + 5 * MMPC_WAVE));
        err_p = MMPC_GRED_MAXERR(RED + 1 * MMPC_WAVE);
        tzmax = MMPC_GRED_MAX(RED + 2 * MMPC_WAVE);
        tzmin = MMPC_GRED_MIN(RED + 3 * MMPC_WAVE);
        // IPOPT's termination test (Waechter & Biegler 2006, eq. 5-6): stationarity over s_d (all multipliers), complementarity
        // over s_c (the row multipliers alone)
        const double zrow = MMPC_GRED_SUM(RED + 4 * MMPC_WAVE);
        zsum = zrow + MMPC_GRED_SUM(HXX);
        double sd = zsum / (double)(nrows_act + NS * NX), sc = zrow / (double)(nrows_act > 0 ? nrows_act : 1);
        sd = (sd > 100.0 ? sd : 100.0) / 100.0;
        sc = (sc > 100.0 ? sc : 100.0) / 100.0;
        E0 = mmpc_max_err(mmpc_max_err(err_d / sd, err_p), tzmax / sc);
        if (!mmpc_finite(E0) || !mmpc_finite(zsum)) {   // (NaN fails the comparison inside mmpc_finite too)
#ifdef MMPC_EMU_DEBUG
            fprintf(stderr, "generic: E0 not finite at it %d: err_d %g err_p %g tzmax %g zsum %g\n", it, err_d, err_p, tzmax, zsum);
#endif
            status = 2; break; }
        if (E0 <= tol) { status = 0; break; }
        if (it == P.max_iter) break;
        {
            // barrier schedule (Fiacco-McCormick, as IPOPT's monotone mode)
            bool changed = false;
            for (;;) {
                const double compmu = mmpc_max(fabs(tzmax - mu), fabs(tzmin - mu));
                const double Emu = mmpc_max(mmpc_max(err_d / sd, err_p), compmu / sc);
                if (!(Emu <= 10.0 * mu && mu > tol / 10)) break;
                mu = mmpc_max(tol / 10, mmpc_min(0.2 * mu, mu * sqrt(mu)));
                changed = true;
            }
            if (changed) filt_init = 0;
        }

        MMPC_GS(1)
        // ============================================================ Newton direction
        // Second-order correction (Waechter & Biegler 2006, section 2.4): when the first trial step is rejected without reducing the
        // infeasibility - and theta(x_k) <= theta_min, the regime of the switching condition, where the Maratos effect lives - the
        // direction is recomputed from x_k with c_soc = alpha c(x_k) + c(x_k + alpha d) in place of the constraint residuals (CD, HR),
        // at most MMPC_SOC_MAX times (c_soc <- alpha_soc c_soc + c(x_soc)).  The row multipliers take their step of the uncorrected
        // direction first (it does not depend on alpha; the specialised kernels, which move to a trial point in place, have applied
        // it by then): the corrected system is the primal-dual Newton system at (x_k, z+).  The uncorrected direction waits in this
        // instance's block of global memory (io.soc) and the line search goes on along it when the correction fails.
        // slacks and multipliers of the rows by (a, a_z) along the direction in DX .. DTR (one item per (row, stage), row-major as in
        // D2); the multipliers are safeguarded with the slack at a whether or not the slack itself moves
        auto row_step = [&](double a, double a_z, bool do_t, bool do_z) {
            LANES_BEGIN
            for (int item = lane; item < R * NS; item += MMPC_WAVE) {
                const int r = item / NS, k = item - r * NS;
                {
                    double dtv, b;
                    if (r < SL_C) {
                        if (!box_bound(k, r, b)) continue;
                        double h, jd;
                        if (r < NU) { h = b - U[k * NU + r]; jd = -DU[k * NU + r]; }
                        else if (r < SL_XLO) { h = U[k * NU + r - NU] - b; jd = DU[k * NU + r - NU]; }
                        else if (r < SL_XHI) { h = b - X[k * NX + r - SL_XLO]; jd = -DX[k * NX + r - SL_XLO]; }
                        else { h = X[k * NX + r - SL_XHI] - b; jd = DX[k * NX + r - SL_XHI]; }
                        dtv = -(h + T[k * R + r]) - jd;
                    } else if (r >= SL_Q && k == 0) continue;
                    else dtv = DTR[k * NR + r - SL_C];
                    const double t = T[k * R + r], z = Z[k * R + r], it_ = mmpc_rcp(t);
                    const double dzv = mu * it_ - z - (z * it_) * dtv;      // (the same expression as in D2: ad was formed from it)
                    const double tn = t + a * dtv;
                    if (do_t) T[k * R + r] = tn;
                    if (do_z) Z[k * R + r] = mmpc_z_safeguard(z + a_z * dzv, tn, mu);
                }
            }
            LANES_END
        };
        int soc_p = 0;
        bool z_done = false, fatal = false;
        double alpha = 0.0, ap0 = 1.0, ad0 = 1.0, dphi0 = 0.0, th_prev = 0.0, phi0 = 0.0, th0 = 0.0;
        for (;;) {   // direction passes: the regular one, then the corrected ones
        int failed = 0;
        // Newton matrix: the exact Lagrangian Hessian; on a non-positive pivot of the recursion IPOPT's inertia correction
        // (Waechter & Biegler 2006, Algorithm IC) - + delta_w I on (x, u), delta_w from a geometric sequence (MMPC_IC_D0, x MMPC_IC_UP
        // per failed pass; an iteration after a corrected one starts from MMPC_IC_DN times the last successful value) until every
        // pivot is positive.  (Rounds 1-3 went down a ladder exact -> exact without lam^T d2f -> Gauss-Newton instead: full
        // Gauss-Newton steps that crawl along a saddle of the barrier problem for hundreds of iterations were the tail of the
        // iteration counts.)  With the terminal equality the multipliers of the regularised system grow like delta_w (the full
        // correction E dx_N = e is forced whatever the damping) and feed back into lam^T d2f: there the second and last rung is
        // Gauss-Newton, as before.
        double dw = (!teq && delta_prev > MMPC_IC_SKIP0) ? mmpc_max(1e-20, MMPC_IC_DN * delta_last) : 0.0;
        for (int attempt = 0;;) {
            const bool exact = attempt == 0, dyn_curv = exact;
            const double reg = prox + dw;
            // ---- A1: stage Hessian / gradient assembly
            MMPC_G2()
            LANES_BEGIN
            for (int k = lane; k < NS; k += MMPC_WAVE) {
                double *hxx = HXX + k * NXX, *qx = QX + k * NX;
                mmpc_state_cost<KIND>(WTS, k == N, X + k * NX, XREF + k * NREF, nullptr, hxx, exact);
                for (int j = 0; j < NX; j++) hxx[j * (j + 1) / 2 + j] += reg;
                for (int j = 0; j < NX; j++) qx[j] = GX[k * NX + j];
                if (k < N) {
                    for (int a = 0; a < NU; a++) { HUUD[k * NU + a] = reg; QU[k * NU + a] = GU[k * NU + a]; }
                    double h02 = 0.0;
                    if (dyn_curv) {
                        // - sum_j lam_{k+1,j} d2 f_j/d(x,u)2 : only f3, f4 are nonlinear (base.py:23-24)
                        const double *cv = CV + k * MMPC_NCV;
                        const double l3 = LAM[(k + 1) * NX + 3], l4 = LAM[(k + 1) * NX + 4];
                        hxx[5] += l3 * cv[4] - l4 * cv[3];   // (2,2): dt*u0*(l3 cos + l4 sin)
                        hxx[19] += dt * l3;   // (5,4)
                        hxx[18] -= dt * l4;   // (5,3)
                        h02 = -(-l3 * cv[10] + l4 * cv[9]);
                    }
                    HUX02[k] = h02;
                }
                double hss = 2 * Sw, gss = 2 * Sw * S[k], vx[6] = {0, 0, 0, 0, 0, 0};
                MMPC_GSA(0)
                // The rows of the stage add to REGISTERS - the diagonal terms of the box rows, the 2 x 2 block of the circle rows, the
                // 6 x 6 block over y = (x, y, psi, q1, q2, q3) of the arm's rows with its gradient - and these to the stage's block in LDS
                // once, after each group of rows: every "+=" on the LDS block was a load -> add -> store round trip the compiler could not overlap
                // with the next (the row gradients it reads may alias the block): 570 cycles per box row, 4 k per half-space row.
                double dgx[NX], dqx[NX], dgu[NU > 0 ? NU : 1], dqu[NU > 0 ? NU : 1];
                for (int j = 0; j < NX; j++) { dgx[j] = 0.0; dqx[j] = 0.0; }
                for (int a = 0; a < NU; a++) { dgu[a] = 0.0; dqu[a] = 0.0; }
                if (k < N)
                    for (int a = 0; a < NU; a++) {
                        double b;
                        if (box_bound(k, a, b)) {
                            const double t = T[k * R + a], z = Z[k * R + a], w = z / t, h = b - U[k * NU + a];
                            dgu[a] += w; dqu[a] -= mu / t + w * (h + t);
                        }
                        if (box_bound(k, NU + a, b)) {
                            const double t = T[k * R + NU + a], z = Z[k * R + NU + a], w = z / t, h = U[k * NU + a] - b;
                            dgu[a] += w; dqu[a] += mu / t + w * (h + t);
                        }
                    }
                for (int j = 0; j < NX; j++) {
                    double b;
                    if (box_bound(k, SL_XLO + j, b)) {
                        const double t = T[k * R + SL_XLO + j], z = Z[k * R + SL_XLO + j], w = z / t, h = b - X[k * NX + j];
                        dgx[j] += w; dqx[j] -= mu / t + w * (h + t);
                    }
                    if (box_bound(k, SL_XHI + j, b)) {
                        const double t = T[k * R + SL_XHI + j], z = Z[k * R + SL_XHI + j], w = z / t, h = X[k * NX + j] - b;
                        dgx[j] += w; dqx[j] += mu / t + w * (h + t);
                    }
                }
                if (k < N) for (int a = 0; a < NU; a++) { HUUD[k * NU + a] += dgu[a]; QU[k * NU + a] += dqu[a]; }
                for (int j = 0; j < NX; j++) { hxx[j * (j + 1) / 2 + j] += dgx[j]; qx[j] += dqx[j]; }
                MMPC_GSA(1)
                double c00 = 0.0, c01 = 0.0, c11 = 0.0, cq0 = 0.0, cq1 = 0.0;
                for (int m = 0; m < M; m++) {
                    const double t = T[k * R + SL_C + m], z = Z[k * R + SL_C + m], w = z / t;
                    const double zh = mu / t + w * (HR[k * NR + m] + t);
                    const double g0 = GC[(k * M + m) * 2], g1 = GC[(k * M + m) * 2 + 1];
                    c00 += w * g0 * g0; c01 += w * g1 * g0; c11 += w * g1 * g1;
                    if (exact) {
                        c00 += z * HC[(k * M + m) * 3]; c01 += z * HC[(k * M + m) * 3 + 1];
                        c11 += z * HC[(k * M + m) * 3 + 2];
                    }
                    cq0 += g0 * zh; cq1 += g1 * zh;
                    hss += w; gss -= zh; vx[0] += w * g0; vx[1] += w * g1;
                }
                hxx[0] += c00; hxx[1] += c01; hxx[2] += c11; qx[0] += cq0; qx[1] += cq1;
                double yb[21], qy[6];     // y-block of the arm's rows (packed lower) and their gradient
                for (int e2 = 0; e2 < 21; e2++) yb[e2] = 0.0;
                for (int a = 0; a < 6; a++) qy[a] = 0.0;
                double hssN = 0.0, gssN = 0.0, vN[6] = {0, 0, 0, 0, 0, 0};
                for (int i = 0; i < NSELF; i++) {
                    const double t = T[k * R + SL_S + i], z = Z[k * R + SL_S + i], w = z / t;
                    const double zh = mu / t + w * (HR[k * NR + M + i] + t);
                    double g6[6];
                    for (int a = 0; a < 6; a++) g6[a] = GSF[(k * NSELF + i) * 6 + a];
                    for (int a = 0; a < 6; a++) {
                        for (int b = 0; b <= a; b++) yb[a * (a + 1) / 2 + b] += w * g6[a] * g6[b];
                        qy[a] += g6[a] * zh;
                    }
                    if (k < N) { hss += w; gss -= zh; for (int a = 0; a < 6; a++) vx[a] += w * g6[a]; }
                    else { hssN += w; gssN -= zh; for (int a = 0; a < 6; a++) vN[a] += w * g6[a]; }
                }
                MMPC_GSA(2)
                // forward kinematics of the stage for the exact curvature of the half-space rows (both forms): once per stage
                double snF = 0.0, csF = 1.0, drF[3] = {0, 0, 0}, dzF[3] = {0, 0, 0};
                if (exact && NHS > 0) {
                    const double *xk = X + k * NX;
                    MMPC_SINCOS(xk[2], &snF, &csF);
                    mmpc_arm_segments(xk[NX - 3], xk[NX - 2], xk[NX - 1], drF, dzF);
                }
                for (int i = 0; i < NHS; i++) {
                    const double t = T[k * R + SL_H + i], z = Z[k * R + SL_H + i], w = z / t;
                    const double zh = mu / t + w * (HR[k * NR + M + NSELF + i] + t);
                    double g6[6];
                    for (int a = 0; a < 6; a++) g6[a] = GHS[(k * 6 + i) * 6 + a];
                    for (int a = 0; a < 6; a++) {
                        for (int b = 0; b <= a; b++) yb[a * (a + 1) / 2 + b] += w * g6[a] * g6[b];
                        qy[a] += g6[a] * zh;
                        vx[a] += w * g6[a];
                    }
                    if (exact) {
                        // curvature z * d2(n.FK point) over (psi, q1, q2, q3): the row multipliers are large (S = 1e5), without
                        // it the step is a Gauss-Newton step and convergence is linear
                        const double *xk = X + k * NX;
                        double gt[6], h10[10];
                        mmpc_hs_row(P, i, xk[0], xk[1], csF, snF, drF, dzF, gt, h10);
                        for (int a = 0; a < 4; a++)
                            for (int b = 0; b <= a; b++) yb[(2 + a) * (3 + a) / 2 + 2 + b] += z * h10[a * (a + 1) / 2 + b];
                    }
                    hss += w; gss -= zh;
                }
                if (NSELF)
                    for (int a = 0; a < 6; a++) {
                        const int ia = kY[a];
                        for (int b = 0; b <= a; b++) hxx[ia * (ia + 1) / 2 + kY[b]] += yb[a * (a + 1) / 2 + b];
                        qx[ia] += qy[a];
                    }
                MMPC_GSA(3)
                if (NQ) {
                    // rows of the NLP as written: a row whose previous-stage entry attains the max acts on x_{k-1}; its blocks are
                    // handed to stage k-1 (HQX, QQX: added in A2), its tie between s_k and x_{k-1} is VQ
                    double hq[21], qq[6] = {0, 0, 0, 0, 0, 0}, vq[6] = {0, 0, 0, 0, 0, 0};
                    for (int e2 = 0; e2 < 21; e2++) { hq[e2] = 0.0; yb[e2] = 0.0; }     // (yb, qy: the rows of this set that act on x_k)
                    for (int a = 0; a < 6; a++) qy[a] = 0.0;
                    if (k >= 1) {
                        const double *xk = X + k * NX, *xp = X + (k - 1) * NX;
                        double sn, cs, dr[3], dz[3], snp, csp, drp[3], dzp[3];
                        if (exact) {
                            sn = snF; cs = csF;
                            for (int a = 0; a < 3; a++) { dr[a] = drF[a]; dz[a] = dzF[a]; }
                            MMPC_SINCOS(xp[2], &snp, &csp); mmpc_arm_segments(xp[NX - 3], xp[NX - 2], xp[NX - 1], drp, dzp);
                        }
                        for (int e = 0; e < NQ; e++) {
                            const double t = T[k * R + SL_Q + e], z = Z[k * R + SL_Q + e], w = z / t;
                            const double zh = mu / t + w * (HR[k * NR + M + NSELF + NHS + e] + t);
                            double g6[6];
                            for (int a = 0; a < 6; a++) g6[a] = GQ8[(k * NQ + e) * 6 + a];
                            const bool br = BQ8[k * NQ + e] != 0.0;
                            double h10[10];
                            if (exact) { double gt[6]; int b2; q8_row(e, xk, cs, sn, dr, dz, xp, csp, snp, drp, dzp, b2, gt, h10); }
                            if (br) {
                                for (int a = 0; a < 6; a++) {
                                    for (int b = 0; b <= a; b++) hq[a * (a + 1) / 2 + b] += w * g6[a] * g6[b];
                                    qq[a] += g6[a] * zh; vq[a] += w * g6[a];
                                }
                                if (exact) for (int a = 0; a < 4; a++) for (int b = 0; b <= a; b++) hq[(2 + a) * (3 + a) / 2 + 2 + b] += z * h10[a * (a + 1) / 2 + b];
                            } else {
                                for (int a = 0; a < 6; a++) {
                                    for (int b = 0; b <= a; b++) yb[a * (a + 1) / 2 + b] += w * g6[a] * g6[b];
                                    qy[a] += g6[a] * zh; vx[a] += w * g6[a];
                                }
                                if (exact) for (int a = 0; a < 4; a++) for (int b = 0; b <= a; b++) yb[(2 + a) * (3 + a) / 2 + 2 + b] += z * h10[a * (a + 1) / 2 + b];
                            }
                            hss += w; gss -= zh;
                        }
                    }
                    for (int a = 0; a < 6; a++) {
                        const int ia = kY[a];
                        for (int b = 0; b <= a; b++) hxx[ia * (ia + 1) / 2 + kY[b]] += yb[a * (a + 1) / 2 + b];
                        qx[ia] += qy[a];
                    }
                    for (int e2 = 0; e2 < 21; e2++) HQX[k * 21 + e2] = hq[e2];
                    for (int a = 0; a < 6; a++) { QQX[k * 6 + a] = qq[a]; VQ[k * 6 + a] = vq[a]; }
                    if (k == N) { for (int e2 = 0; e2 < 21; e2++) HQX[(N + 1) * 21 + e2] = 0.0; for (int a = 0; a < 6; a++) QQX[(N + 1) * 6 + a] = 0.0; }
                }
                HSS[k] = hss; GSS[k] = gss;
                if (NQ && k == N - 1) { MISC[4] = hss; MISC[5] = gss; }   // (copies for the lane of stage N-2, see A2)
                for (int a = 0; a < 6; a++) VX[k * 6 + a] = vx[a];
                if (k == N) { MISC[2] = hssN; MISC[3] = gssN; for (int a = 0; a < 6; a++) VXN[a] = vN[a]; }
                MMPC_GSA(4)
            }
            LANES_END
            // ---- A2: Schur complement of s_k (H_xs = -v); stage N-1 also carries the terminal
            //          self rows (Q1): a = v + A^T vN, b = B^T vN, gamma = g_s - vN.c
            LANES_BEGIN
            // s_{N-1} reaching back to x_{N-2} touches three stages (x_{N-2}, x_{N-1} and - quirk Q1 - x_N): no stage-wise
            // elimination of it is exact.  It then stays a BORDER variable sigma of the stage-wise system,
            //   [K -v~; -v~^T h] [d; dsigma] = -[q; g_s],   v~ = (vq on x_{N-2}, v on x_{N-1}, vN on x_N),
            // solved with one more right-hand side through the same recursion (column 2 of PNU / GNU / KFV, beside the two
            // multipliers of the terminal equality) and a scalar pivot h - v~.K^{-1} v~.  (Round 2 kept the diagonal block of x_{N-2}
            // only - an inexact Newton matrix in that corner: 2 % of 2048 starts around the demo's planes stalled at a dual
            // residual of 1e-2 .. 1e-6 until the iteration cap, the others needed 82 iterations on average instead of 30.)
            bool sig = false;
            if (NQ && N >= 2) for (int a = 0; a < 6; a++) sig = sig || VQ[(N - 1) * 6 + a] != 0.0;
            if (lane == 0) SIGW[0] = sig ? 1.0 : 0.0;
            for (int k = lane; k < NS; k += MMPC_WAVE) {
                double *hxx = HXX + k * NXX, *qx = QX + k * NX;
                bool back_self = false;   // s_k reaches back to x_{k-1}: eliminated at stage k-1, not here
                if (NQ) {
                    // blocks the rows of stage k+1 contribute to this stage
                    for (int a = 0; a < 6; a++) {
                        const int ia = kY[a];
                        for (int b = 0; b <= a; b++) hxx[ia * (ia + 1) / 2 + kY[b]] += HQX[(k + 1) * 21 + a * (a + 1) / 2 + b];
                        qx[ia] += QQX[(k + 1) * 6 + a];
                    }
                    if (k >= 1 && k != N - 1) for (int a = 0; a < 6; a++) back_self = back_self || VQ[k * 6 + a] != 0.0;
                    bool back_next = false;
                    if (k + 1 <= N && k + 1 != N - 1) for (int a = 0; a < 6; a++) back_next = back_next || VQ[(k + 1) * 6 + a] != 0.0;
                    if (k < N) {
                        for (int c = 0; c < NU * NX; c++) HUXS[k * NU * NX + c] = 0.0;
                        for (int c = 0; c < NUU; c++) HUUS[k * NUU + c] = 0.0;
                    }
                    if (back_next) {
                        // Schur complement of s_{k+1} in the variables of this stage: dx_{k+1} = A dx + B du + c, so
                        // a = vq + A^T v, b = B^T v, gamma = g_s - v.c  (the pattern of the terminal self rows, Q1)
                        const double ih = 1.0 / HSS[k + 1];
                        double gam = GSS[k + 1], vfull[NX], a[NX], b[NU > 0 ? NU : 1];
                        for (int j = 0; j < NX; j++) { vfull[j] = 0.0; a[j] = 0.0; }
                        for (int q = 0; q < 6; q++) { vfull[kY[q]] = VX[(k + 1) * 6 + q]; a[kY[q]] = VQ[(k + 1) * 6 + q]; }
                        const double *cv = CV + k * MMPC_NCV;
                        for (int j = 0; j < NV; j++) {
                            double v = 0.0;
                            for (int q = 0; q < 4; q++) v += cv[TB::ccv(j, q)] * vfull[TB::crow(j, q)];
                            if (j < NX) a[j] += v; else b[j - NX] = v;
                        }
                        for (int j = 0; j < NX; j++) gam -= vfull[j] * CD[k * NX + j];
                        for (int e = 0; e < NXX; e++) hxx[e] -= a[kTriI[e]] * a[kTriJ[e]] * ih;
                        for (int j = 0; j < NX; j++) qx[j] += a[j] * gam * ih;
                        for (int c = 0; c < NU; c++) {
                            for (int j = 0; j < NX; j++) HUXS[(k * NU + c) * NX + j] = -b[c] * a[j] * ih;
                            for (int d = 0; d <= c; d++) HUUS[k * NUU + c * (c + 1) / 2 + d] = -b[c] * b[d] * ih;
                            QU[k * NU + c] += b[c] * gam * ih;
                        }
                    }
                }
                if (back_self) {
                    // (nothing: the slack of this stage has been eliminated by the lane of stage k-1 above)
                } else if (k == N - 1 && NSELF && sig) {
                    // border variable: totals of its row block for the border solve and D1, no Schur complement
                    HSS[k] = HSS[k] + MISC[2]; GSS[k] = GSS[k] + MISC[3];
                    for (int c = 0; c < NU * NX; c++) HUXL[c] = 0.0;
                    for (int c = 0; c < NUU; c++) HUUL[c] = 0.0;
                } else if (k == N - 1 && NSELF) {
                    const double hss = HSS[k] + MISC[2];
                    double gam = GSS[k] + MISC[3];
                    HSS[k] = hss; GSS[k] = gam;
                    const double ih = 1.0 / hss;
                    double vfull[NX], a[NX], b[NU > 0 ? NU : 1];
                    for (int j = 0; j < NX; j++) { vfull[j] = 0.0; a[j] = 0.0; }
                    for (int q = 0; q < 6; q++) { vfull[kY[q]] = VXN[q]; a[kY[q]] = VX[k * 6 + q]; }
                    const double *cv = CV + k * MMPC_NCV;
                    for (int j = 0; j < NV; j++) {
                        double v = 0.0;
                        for (int q = 0; q < 4; q++) v += cv[TB::ccv(j, q)] * vfull[TB::crow(j, q)];
                        if (j < NX) a[j] += v; else b[j - NX] = v;
                    }
                    for (int j = 0; j < NX; j++) gam -= vfull[j] * CD[k * NX + j];
                    for (int e = 0; e < NXX; e++) hxx[e] -= a[kTriI[e]] * a[kTriJ[e]] * ih;
                    for (int j = 0; j < NX; j++) { qx[j] += a[j] * gam * ih; VXN[6 + j] = a[j]; }
                    for (int c = 0; c < NU; c++) {
                        for (int j = 0; j < NX; j++) HUXL[c * NX + j] = -b[c] * a[j] * ih;
                        for (int d = 0; d <= c; d++) HUUL[c * (c + 1) / 2 + d] = -b[c] * b[d] * ih;
                        QU[k * NU + c] += b[c] * gam * ih;
                    }
                } else {
                    if (k == N - 1) {  // no self rows: still define the dense last-stage blocks
                        for (int c = 0; c < NU * NX; c++) HUXL[c] = 0.0;
                        for (int c = 0; c < NUU; c++) HUUL[c] = 0.0;
                    }
                    const double ih = 1.0 / HSS[k], gam = GSS[k];
                    const double *v = VX + k * 6;
                    const int ny = NSELF ? 6 : 2;
                    for (int a = 0; a < ny; a++) {
                        const int ia = kY[a];
                        for (int b = 0; b <= a; b++) hxx[ia * (ia + 1) / 2 + kY[b]] -= v[a] * v[b] * ih;
                        qx[ia] += v[a] * gam * ih;
                    }
                }
            }
            // unpack P_N (stage-N Hessian after its Schur step is written by another lane in this
            // phase, so the unpack happens in R0 below)
            LANES_END
            MMPC_GSA(5)
            MMPC_GS(2)
            // ---- R0: full copy of P_N
            const bool sig = NQ && SIGW[0] != 0.0;       // (uniform; written by A2)
            const bool brd = teq || sig;                  // border columns in play
            // y-entry of a state index (x, y, psi, q1, q2, q3 <-> 0..5), -1: none
            auto yix = [&](int i) -> int { return i < 3 ? i : (i >= NX - 3 ? i - (NX - 6) : -1); };
#if MMPC_GEN_TILE
            {
                // entry (ti, tj) of stage k's matrix [Hxx Hxu q_x; Hux Huu q_u; q^T 0] in tile numbering (what R2 of the scalar
                // pass adds to [A B]^T P [A B]): packed Hxx; Hux = (0,2) entry of the dynamics curvature + the dense blocks of a
                // slack eliminated here (last stage: quirk Q1; as-written rows: any stage); Huu = R2 + W2 + diagonal barrier terms
                // + the same dense blocks; gradient in row / column NX
                auto stage_entry = [&](const unsigned (&t)[4], const int k) -> double {
                    double x[4];
                    for (int q = 0; q < 4; q++) {
                        const unsigned w = t[q];
                        const double v = lds[(int)(w & 0xffffu) + k * (int)((w >> 16) & 0x7fffu)];
                        x[q] = ((w >> 31) && k != N - 1) ? 0.0 : v;
                    }
                    return ((x[0] + x[1]) + x[2]) + x[3];
                };
                // R0: terminal cost-to-go [P_N p_N; p_N^T .] in accumulator layout (+ the augmentation of the terminal equality),
                //     operands of stage N-1
                MMPC_G2()
                LANES_BEGIN
                {
                    auto &ls = MMPC_LS;
                    const int g = lane >> 4, j = lane & 15;
                    for (int r = 0; r < 4; r++) {
                        int i = tvar(g + 4 * r), jj = tvar(j);
                        if (i < jj) { const int t_ = i; i = jj; jj = t_; }
                        double v = 0.0;
                        if (i < NX) v = HXX[N * NXX + i * (i + 1) / 2 + jj] + ((teq && i == jj && i < 2) ? MMPC_RHO_EQ : 0.0);
                        else if (i == NV && jj < NX) v = QX[N * NX + jj] - ((teq && jj < 2) ? MMPC_RHO_EQ * (XREF[N * NX + jj] - X[N * NX + jj]) : 0.0);
                        ls.rP[r] = v;
                        ls.rM[r] = stage_entry(ls.st_o[r], N - 1);
                    }
                    for (int r = 0; r < NKB; r++) { const unsigned o = ls.ab_o[r]; ls.rAB[r] = lds[(o & 0xffffu) + (N - 1) * (int)(o >> 16)]; }
                }
                if (brd)
                    for (int e = lane; e < NX * MMPC_NBC; e += MMPC_WAVE) {   // p_N of the border columns: E^T (E = [I2 0]) and -vN
                        const int i = e / MMPC_NBC, c = e % MMPC_NBC;
                        double v = 0.0;
                        if (c < 2) v = (teq && i == c) ? 1.0 : 0.0;
                        else if (sig && yix(i) >= 0) v = -VXN[yix(i)];
                        PNU[e] = v; PNUS[N * NX * MMPC_NBC + e] = v;
                    }
                LANES_END
                int ric_bad = 0;
                MMPC_GS2(0)
                // (two stages per trip where the horizon is a constant of the instantiation - the static shapes -, as in the specialised
                //  kernel: measured 1 / 2 / 4: C4 batch 32.1 / 30.6 / 31.2 ms, demo shape as written 48.8 / 41.9 / 41.6 ms)
                constexpr int RICU = NC > 0 ? MMPC_GEN_RIC_UNROLL : 1;
#pragma unroll RICU
                for (int k = N - 1; k >= 0; k--) {
                    // R1: T = [P p; p^T .] [A B c; 0 0 1];  R2: M = [A B c; 0 0 1]^T T + stage matrix
                    MMPC_MFMA0(rT, ls.rP[0], ls.rAB[0])
                    MMPC_MFMA(rT, ls.rP[1], ls.rAB[1])
                    if (NKB > 2) MMPC_MFMA(rT, ls.rP[NKB > 2 ? 2 : 0], ls.rAB[NKB > 2 ? 2 : 0])
                    MMPC_MFMA(rM, ls.rAB[0], ls.rT[0])
                    MMPC_MFMA(rM, ls.rAB[1], ls.rT[1])
                    if (NKB > 2) MMPC_MFMA(rM, ls.rAB[NKB > 2 ? 2 : 0], ls.rT[NKB > 2 ? 2 : 0])
                    MMPC_GS2(1)
                    // operands of the next stage travel while this one eliminates its inputs
                    LANES_BEGIN
                    {
                        auto &ls = MMPC_LS;
                        if (k > 0) {
                            for (int r = 0; r < NKB; r++) { const unsigned o = ls.ab_o[r]; ls.nab[r] = lds[(o & 0xffffu) + (k - 1) * (int)(o >> 16)]; }
                            for (int r = 0; r < 4; r++) ls.nhm[r] = stage_entry(ls.st_o[r], k - 1);
                        }
                    }
                    LANES_END_REG
                    MMPC_GS2(2)
                    // R3: the inputs are eliminated on the tile by rank-one MFMAs (mmpc_fast.h: pairs of inputs whose rows share an
                    // accumulator register go as two K-slots of one MFMA, in sequential L D L^T arithmetic); the normalised pivot
                    // rows go to KK / KF / KU, 1 / pivot to PIV (for the border columns)
                    for (int leg = 0; leg < NLEG; leg++) {
                        LANES_BEGIN
                        {
                            auto &ls = MMPC_LS;
                            const int a0 = PAIRS ? 2 * leg : leg;
                            const bool pair = PAIRS && a0 + 1 < NU;
                            const int ta = NX + 1 + a0, ra = ta >> 2, ga = ta & 3, g = lane >> 4;
                            const double c = ls.rM[ra];
                            double w, cb = c;
                            bool own;
                            if (pair) {
                                const double d00 = MMPC_LANE_GET(rM[ra], 16 * ga + ta), d01 = MMPC_LANE_GET(rM[ra], 16 * ga + ta + 1),
                                             d11 = MMPC_LANE_GET(rM[ra], 16 * (ga + 1) + ta + 1);
                                const double i0 = mmpc_rcp3(d00), l = d01 * i0, d1 = fma(-l, d01, d11);
                                if (!(d00 > 0.0 && d1 > 0.0)) ric_bad = 1;
                                const double i1 = mmpc_rcp3(d1), co = MMPC_LANE_XOR16(rM[ra]);
                                const bool in1 = g == ga + 1;
                                own = g == ga || in1;
                                cb = fma(in1 ? -l : 0.0, co, c);      // (every lane uses the exchanged value, with a zero multiplier where it does not apply)
                                w = cb * (in1 ? i1 : i0);
                                if (lane == 0) { PIV[k * NU + a0] = i0; PIV[k * NU + a0 + 1] = i1; }
                            } else {
                                const double d = MMPC_LANE_GET(rM[ra], 16 * ga + ta);
                                if (!(d > 0.0)) ric_bad = 1;
                                own = g == ga;
                                const double id = mmpc_rcp3(d);
                                w = c * id;
                                if (lane == 0) PIV[k * NU + a0] = id;
                            }
                            ls.opa = own ? -w : 0.0;
                            ls.opb = own ? cb : 0.0;
                            lds[ls.kl_b[leg] + k * ls.kl_s[leg]] = w;
                        }
                        LANES_END_REG
                        MMPC_MFMA(rM, ls.opa, ls.opb)
                    }
                    MMPC_GS2(3)
                    LANES_BEGIN
                    {
                        auto &ls = MMPC_LS;
                        for (int r = 0; r < 4; r++) {
                            ls.rP[r] = ls.rM[r];
                            lds[ls.p_o[r] + k * ls.p_s[r]] = ls.rM[r];
                            ls.rM[r] = ls.nhm[r];
                        }
                        for (int r = 0; r < NKB; r++) ls.rAB[r] = ls.nab[r];
                    }
                    LANES_END
                    MMPC_GS2(4)
                    if (ric_bad) { failed = 1; break; }
                }
                if (!failed) {
                    // gains from the normalised pivot rows by back-substitution over the inputs: K_a = -(w_a[x,1] + sum_{b>a} w_a[u_b] K_b)
                    LANES_BEGIN
                    for (int i = lane; i < N * (NX + 1); i += MMPC_WAVE) {
                        const int kk = i / (NX + 1), jj = i % (NX + 1);
                        double kv[NU > 0 ? NU : 1];
                        for (int a = 0; a < NU; a++) kv[a] = jj < NX ? KK[(kk * NU + a) * NX + jj] : KF[kk * NU + a];
                        for (int a = NU - 1; a >= 0; a--) {
                            double v = kv[a];
                            for (int b2 = a + 1; b2 < NU; b2++) v += KU[kk * NPU + a * (2 * NU - a - 1) / 2 + (b2 - a - 1)] * kv[b2];
                            kv[a] = -v;
                        }
                        for (int a = 0; a < NU; a++) { if (jj < NX) KK[(kk * NU + a) * NX + jj] = kv[a]; else KF[kk * NU + a] = kv[a]; }
                    }
                    LANES_END
                    if (brd) {
                        // border columns through the same factorisation: with bp = B^T p_{k+1},  kf = -Hh^{-1} bp  (Hh = L D L^T: L from
                        // the couplings KU, 1 / D in PIV)  and  p_k = A^T p_{k+1} + K^T bp + (what the column adds at this stage)
                        // (two short phases per stage - bp by one lane per (column, input), p_k by one lane per (column, state), both
                        //  reading p_{k+1} from its stored copy - and the kf of all stages and columns in one phase afterwards: the
                        //  same sums in the same order as one lane per column would form them, a third of the time)
                        double *const BPS = GNU;   // (the scalar pass's array: NV x NBC >= NBC x NU doubles)
                        for (int k = N - 1; k >= 0; k--) {
                            const double *cv = CV + k * MMPC_NCV;
                            const double *pn = PNUS + (k + 1) * NX * MMPC_NBC;
                            LANES_BEGIN
                            if (lane < MMPC_NBC * NU) {
                                const unsigned w = MMPC_LS.bdA;
                                const int c = lane / (NU > 0 ? NU : 1);
                                double v = 0.0;
                                for (int q = 0; q < 4; q++) v += cv[(w >> (8 * q + 4)) & 15u] * pn[((w >> (8 * q)) & 15u) * MMPC_NBC + c];
                                BPS[lane] = v;
                            }
                            LANES_END
                            LANES_BEGIN
                            if (lane < MMPC_NBC * NX) {
                                const unsigned w = MMPC_LS.bdB;
                                const int c = lane / NX, i = lane - c * NX;
                                double v = 0.0;
                                for (int q = 0; q < 4; q++) v += cv[(w >> (8 * q + 4)) & 15u] * pn[((w >> (8 * q)) & 15u) * MMPC_NBC + c];
                                for (int a = 0; a < NU; a++) v += KK[(k * NU + a) * NX + i] * BPS[c * NU + a];
                                if (c == 2 && sig && yix(i) >= 0) {
                                    if (k == N - 1) v -= VX[(N - 1) * 6 + yix(i)];
                                    if (k == N - 2) v -= VQ[(N - 1) * 6 + yix(i)];
                                }
                                PNUS[(k * NX + i) * MMPC_NBC + c] = v;
                            }
                            LANES_END
                        }
                        LANES_BEGIN
                        for (int item = lane; item < N * MMPC_NBC; item += MMPC_WAVE) {
                            const int k = item / MMPC_NBC, c = item - k * MMPC_NBC;
                            const double *cv = CV + k * MMPC_NCV;
                            double pn[NX], y[NU > 0 ? NU : 1];
                            for (int i = 0; i < NX; i++) pn[i] = PNUS[((k + 1) * NX + i) * MMPC_NBC + c];
                            for (int a = 0; a < NU; a++) {
                                double v = 0.0;
                                for (int q = 0; q < 4; q++) v += cv[TB::ccv(NX + a, q)] * pn[TB::crow(NX + a, q)];
                                y[a] = v;
                            }
                            // L y' = bp (unit lower: L[b][a] = KU[k][a, b], b > a), y'' = D^{-1} y', L^T z = y''
                            for (int a = 0; a < NU; a++) for (int b2 = a + 1; b2 < NU; b2++) y[b2] -= KU[k * NPU + a * (2 * NU - a - 1) / 2 + (b2 - a - 1)] * y[a];
                            for (int a = 0; a < NU; a++) y[a] *= PIV[k * NU + a];
                            for (int a = NU - 1; a >= 0; a--) for (int b2 = a + 1; b2 < NU; b2++) y[a] -= KU[k * NPU + a * (2 * NU - a - 1) / 2 + (b2 - a - 1)] * y[b2];
                            for (int a = 0; a < NU; a++) KFV[(k * NU + a) * MMPC_NBC + c] = -y[a];
                        }
                        LANES_END
                    }
                }
                MMPC_GS2(5)
            }
#else
            LANES_BEGIN
            for (int e = lane; e < NX * NX; e += MMPC_WAVE) {
                const int i = e / NX, j = e % NX;
                PF[e] = HXX[N * NXX + (i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i)] + ((teq && i == j && i < 2) ? MMPC_RHO_EQ : 0.0);
            }
            if (brd)
                for (int e = lane; e < NX * MMPC_NBC; e += MMPC_WAVE) {   // p_N of the border columns: E^T (E = [I2 0]) and -vN
                    const int i = e / MMPC_NBC, c = e % MMPC_NBC;
                    double v = 0.0;
                    if (c < 2) v = (teq && i == c) ? 1.0 : 0.0;
                    else if (sig && yix(i) >= 0) v = -VXN[yix(i)];
                    PNU[e] = v; PNUS[N * NX * MMPC_NBC + e] = v;
                }
            LANES_END
            // ---- Riccati recursion (block LDL^T of the stage-wise KKT matrix)
            for (int k = N - 1; k >= 0; k--) {
                const double *cv = CV + k * MMPC_NCV;
                // R1: T = P [A B],  pc = p + P c
                LANES_BEGIN
                for (int e = lane; e < NX * NV + NX; e += MMPC_WAVE) {
                    if (e < NX * NV) {
                        const int i = e / NV, j = e % NV;
                        double v = 0.0;
                        for (int q = 0; q < 4; q++) v += PF[i * NX + TB::crow(j, q)] * cv[TB::ccv(j, q)];
                        TT[e] = v;
                    } else {
                        const int i = e - NX * NV;
                        double v = QX[(k + 1) * NX + i];
                        if (teq && k == N - 1 && i < 2) v -= MMPC_RHO_EQ * (XREF[N * NX + i] - X[N * NX + i]);
                        for (int m = 0; m < NX; m++) v += PF[i * NX + m] * CD[k * NX + m];
                        PC[i] = v;
                    }
                }
                LANES_END
                // R2: [F G^T; G Hh] = [A B]^T T + stage Hessian;  [gx; gu] = q + [A B]^T pc
                LANES_BEGIN
                for (int e = lane; e < NXX + NU * NX + NUU + NV; e += MMPC_WAVE) {
                    if (e < NXX) {
                        const int i = kTriI[e], j = kTriJ[e];
                        double v = HXX[k * NXX + e];
                        for (int q = 0; q < 4; q++) v += cv[TB::ccv(i, q)] * TT[TB::crow(i, q) * NV + j];
                        MF[e] = v;
                    } else if (e < NXX + NU * NX) {
                        const int e2 = e - NXX, a = e2 / NX, j = e2 % NX, col = NX + a;
                        double v = (k == N - 1 ? HUXL[e2] : 0.0) + (NQ ? HUXS[k * NU * NX + e2] : 0.0) + ((a == 0 && j == 2) ? HUX02[k] : 0.0);
                        for (int q = 0; q < 4; q++) v += cv[TB::ccv(col, q)] * TT[TB::crow(col, q) * NV + j];
                        MG[e2] = v;
                    } else if (e < NXX + NU * NX + NUU) {
                        const int e2 = e - NXX - NU * NX, a = kTriI[e2], b = kTriJ[e2], col = NX + a;
                        double v = WTS[MMPC_W_RW2 + a * NU + b] + (a == b ? HUUD[k * NU + a] : 0.0) + (k == N - 1 ? HUUL[e2] : 0.0) + (NQ ? HUUS[k * NUU + e2] : 0.0);
                        for (int q = 0; q < 4; q++) v += cv[TB::ccv(col, q)] * TT[TB::crow(col, q) * NV + NX + b];
                        MH[e2] = v;
                    } else {
                        const int j = e - NXX - NU * NX - NUU;
                        double v = j < NX ? QX[k * NX + j] : QU[k * NU + j - NX];
                        for (int q = 0; q < 4; q++) v += cv[TB::ccv(j, q)] * PC[TB::crow(j, q)];
                        if (j < NX) MGX[j] = v; else MGU[j - NX] = v;
                    }
                }
                if (brd)
                    for (int e = lane; e < NV * MMPC_NBC; e += MMPC_WAVE) {   // g = [A B]^T p_{k+1} of the border columns (+ -v~ of this stage)
                        const int j = e / MMPC_NBC, c = e % MMPC_NBC;
                        double v = 0.0;
                        for (int q = 0; q < 4; q++) v += cv[TB::ccv(j, q)] * PNU[TB::crow(j, q) * MMPC_NBC + c];
                        if (c == 2 && sig && j < NX && yix(j) >= 0) {
                            if (k == N - 1) v -= VX[(N - 1) * 6 + yix(j)];
                            if (k == N - 2) v -= VQ[(N - 1) * 6 + yix(j)];
                        }
                        GNU[e] = v;
                    }
                LANES_END
                // R3/R4: Cholesky of Hh (every solving lane redundantly, in registers), then one
                //        right-hand side per lane: K = -Hh^{-1} G (NX columns), kf = -Hh^{-1} gu
                LANES_BEGIN
                if (lane <= NX + (brd ? MMPC_NBC : 0)) {
                    double Lc[NUU];
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < NU; j++) {
                        double d = MH[j * (j + 1) / 2 + j];
#pragma unroll
                        for (int q = 0; q < j; q++) d -= Lc[j * (j + 1) / 2 + q] * Lc[j * (j + 1) / 2 + q];
                        if (!(d > 0.0) || !mmpc_finite(d)) { ok = false; d = 1.0; }
                        const double ljj = sqrt(d), il = 1.0 / ljj;
                        Lc[j * (j + 1) / 2 + j] = il;  // store the inverse pivot
#pragma unroll
                        for (int i = j + 1; i < NU; i++) {
                            double v = MH[i * (i + 1) / 2 + j];
#pragma unroll
                            for (int q = 0; q < j; q++) v -= Lc[i * (i + 1) / 2 + q] * Lc[j * (j + 1) / 2 + q];
                            Lc[i * (i + 1) / 2 + j] = v * il;
                        }
                    }
                    double rhs[NU];
#pragma unroll
                    for (int a = 0; a < NU; a++) rhs[a] = lane < NX ? MG[a * NX + lane] : (lane == NX ? MGU[a] : GNU[(NX + a) * MMPC_NBC + lane - NX - 1]);
#pragma unroll
                    for (int i = 0; i < NU; i++) {
                        double v = rhs[i];
#pragma unroll
                        for (int q = 0; q < i; q++) v -= Lc[i * (i + 1) / 2 + q] * rhs[q];
                        rhs[i] = v * Lc[i * (i + 1) / 2 + i];
                    }
#pragma unroll
                    for (int i = NU - 1; i >= 0; i--) {
                        double v = rhs[i];
#pragma unroll
                        for (int q = i + 1; q < NU; q++) v -= Lc[q * (q + 1) / 2 + i] * rhs[q];
                        rhs[i] = v * Lc[i * (i + 1) / 2 + i];
                    }
#pragma unroll
                    for (int a = 0; a < NU; a++) {
                        if (lane < NX) KK[(k * NU + a) * NX + lane] = -rhs[a];
                        else if (lane == NX) KF[k * NU + a] = -rhs[a];
                        else KFV[(k * NU + a) * MMPC_NBC + lane - NX - 1] = -rhs[a];
                    }
                    if (!ok && lane == 0) MISC[0] = 1.0;
                }
                LANES_END
                if (MISC[0] != 0.0) { failed = 1; break; }
                // R5: P_k = F + G^T K,  p_k = gx + G^T kf   (overwrite the stage blocks)
                LANES_BEGIN
                if (brd)
                    for (int e = lane; e < NX * MMPC_NBC; e += MMPC_WAVE) {   // p_k = g_x + G^T kf of the border columns
                        const int i = e / MMPC_NBC, c = e % MMPC_NBC;
                        double v = GNU[i * MMPC_NBC + c];
                        for (int a = 0; a < NU; a++) v += MG[a * NX + i] * KFV[(k * NU + a) * MMPC_NBC + c];
                        PNU[e] = v; PNUS[k * NX * MMPC_NBC + e] = v;
                    }
                for (int e = lane; e < NXX + NX; e += MMPC_WAVE) {
                    if (e < NXX) {
                        const int i = kTriI[e], j = kTriJ[e];
                        double v = MF[e];
                        for (int a = 0; a < NU; a++) v += MG[a * NX + i] * KK[(k * NU + a) * NX + j];
                        HXX[k * NXX + e] = v;
                        PF[i * NX + j] = v; PF[j * NX + i] = v;
                    } else {
                        const int i = e - NXX;
                        double v = MGX[i];
                        for (int a = 0; a < NU; a++) v += MG[a * NX + i] * KF[k * NU + a];
                        QX[k * NX + i] = v;
                    }
                }
                LANES_END
            }
#endif
            if (!failed && brd) {
                // the direction is affine in the border variables y = (nu0, nu1, dsigma): roll out the y = 0 solution and the
                // sensitivities (same gains K; homogeneous dynamics for the sensitivities), and collect v~.d of every column
                LANES_BEGIN
                for (int j = lane; j < 2 * (1 + MMPC_NBC) * NX; j += MMPC_WAVE) FWV[j] = 0.0;
                if (lane < 1 + MMPC_NBC) SIGW[4 + lane] = 0.0;
                LANES_END
                for (int k = 0; k < N; k++) {
                    const double *src = FWV + (k & 1) * (1 + MMPC_NBC) * NX;
                    double *dst = FWV + ((k + 1) & 1) * (1 + MMPC_NBC) * NX;
                    LANES_BEGIN
                    if (lane < (1 + MMPC_NBC) * NX) {
                        const int c = lane / NX, i = lane % NX;
                        const double *cv = CV + k * MMPC_NCV;
                        const double *d = src + c * NX;
                        double v = c == 0 ? CD[k * NX + i] : 0.0;
#if MMPC_GEN_TILE
                        {   // (row i of [A B] from the two packed words of the lane: state terms, then the input term - the tables' order)
                            const unsigned st = MMPC_LS.fwb_st, fi = MMPC_LS.fwb_in;
                            const int a = (int)(fi & 255u) - 1;
                            for (int q = 0; q < 4; q++) v += cv[(st >> (8 * q + 4)) & 15u] * d[(st >> (8 * q)) & 15u];
                            if (a >= 0) {
                                double x = c == 0 ? KF[k * NU + a] : KFV[(k * NU + a) * MMPC_NBC + c - 1];
                                for (int j = 0; j < NX; j++) x += KK[(k * NU + a) * NX + j] * d[j];
                                v += cv[(fi >> 8) & 255u] * x;
                            }
                        }
#else
                        for (int q = 0; q < 5; q++) {
                            const int col = TB::rcol(i, q);
                            double x;
                            if (col < NX) x = d[col];
                            else {
                                const int a = col - NX;
                                x = c == 0 ? KF[k * NU + a] : KFV[(k * NU + a) * MMPC_NBC + c - 1];
                                for (int j = 0; j < NX; j++) x += KK[(k * NU + a) * NX + j] * d[j];
                            }
                            v += cv[TB::rcv(i, q)] * x;
                        }
#endif
                        dst[c * NX + i] = v;
                    }
                    LANES_END
                    if (sig && k + 1 >= N - 2) {   // v~ . dx_{k+1}: vq on x_{N-2}, v on x_{N-1}, vN on x_N
                        LANES_BEGIN
                        if (lane < 1 + MMPC_NBC) {
                            const double *vv = k + 1 == N - 2 ? VQ + (N - 1) * 6 : (k + 1 == N - 1 ? VX + (N - 1) * 6 : VXN);
                            double t = SIGW[4 + lane];
                            for (int a = 0; a < 6; a++) t += vv[a] * dst[lane * NX + kY[a]];
                            SIGW[4 + lane] = t;
                        }
                        LANES_END
                    }
                }
                if (sig && !(HSS[N - 1] - SIGW[4 + MMPC_NBC] > 0.0)) failed = 1;   // pivot of the border variable (uniform)
            }
#ifdef MMPC_EMU_DEBUG
            if (failed) fprintf(stderr, "generic it %d: attempt %d lost a pivot (prox %g)\n", it, attempt, prox);
#endif
            if (!failed) { if (dw > 0.0) delta_last = dw; if (soc_p == 0) delta_prev = dw; break; }
            if (teq) { if (attempt >= 1) break; attempt = 1; }
            else {
                dw = dw == 0.0 ? (delta_last == 0.0 ? MMPC_IC_D0 : mmpc_max(1e-20, MMPC_IC_DN * delta_last)) : MMPC_IC_UP * dw;
                if (dw > 1e40) break;
            }
            failed = 0;
            LANES_BEGIN
            if (lane == 0) MISC[0] = 0.0;
            LANES_END
        }
        if (failed) {
#ifdef MMPC_EMU_DEBUG
            fprintf(stderr, "generic: factorisation failed at it %d (prox %g)\n", it, prox);
#endif
            status = 2; fatal = true; break; }

        {
            const bool sig = NQ && SIGW[0] != 0.0, brd = teq || sig;
            if (brd) {
                // the small system in y = (nu0, nu1, dsigma):
                //   E (dx_N + D y) = e                        (terminal equality; D = sensitivities of dx_N)
                //   h dsigma - v~.(d + D y) = -g_s            (stationarity in s_{N-1})
                // rows / columns of variables that are not in play are replaced by the identity
                const double *fin = FWV + (N & 1) * (1 + MMPC_NBC) * NX;
                double Ms[3][4];
                for (int r = 0; r < 2; r++) {
                    for (int c = 0; c < 3; c++) Ms[r][c] = fin[(1 + c) * NX + r];
                    Ms[r][3] = XREF[N * NX + r] - X[N * NX + r] - fin[r];
                }
                Ms[2][0] = -SIGW[5]; Ms[2][1] = -SIGW[6]; Ms[2][2] = (sig ? HSS[N - 1] : 0.0) - SIGW[7]; Ms[2][3] = -(sig ? GSS[N - 1] : 0.0) + SIGW[4];
                for (int r = 0; r < 3; r++) {
                    if (r < 2 ? teq : sig) continue;
                    for (int q = 0; q < 4; q++) Ms[r][q] = 0.0;
                    for (int q = 0; q < 3; q++) Ms[q][r] = 0.0;
                    Ms[r][r] = 1.0;
                }
                for (int pc = 0; pc < 3; pc++) {   // Gaussian elimination with row pivoting
                    int pr = pc;
                    for (int r = pc + 1; r < 3; r++) if (fabs(Ms[r][pc]) > fabs(Ms[pr][pc])) pr = r;
                    if (pr != pc) for (int q = 0; q < 4; q++) { const double tq = Ms[pr][q]; Ms[pr][q] = Ms[pc][q]; Ms[pc][q] = tq; }
                    for (int r = 0; r < 3; r++) if (r != pc) { const double f = Ms[r][pc] / Ms[pc][pc]; for (int q = pc; q < 4; q++) Ms[r][q] -= f * Ms[pc][q]; }
                }
                const double nu0 = Ms[0][3] / Ms[0][0], nu1 = Ms[1][3] / Ms[1][1], dsg = Ms[2][3] / Ms[2][2];
                LANES_BEGIN
                if (lane == 0) { NUEQ[2] = nu0; NUEQ[3] = nu1; SIGW[1] = dsg; }
                for (int e = lane; e < N * NU; e += MMPC_WAVE) KF[e] += KFV[e * MMPC_NBC] * nu0 + KFV[e * MMPC_NBC + 1] * nu1 + KFV[e * MMPC_NBC + 2] * dsg;
                LANES_END
            }
        }
        MMPC_GS(3)
        // ---- forward roll-out of the linearised dynamics
        LANES_BEGIN
        for (int j = lane; j < NX; j += MMPC_WAVE) DX[j] = 0.0;
        LANES_END
#if MMPC_GEN_TILE
        // (one phase per stage: the lane of row i forms the input step its row needs itself - nine products - instead of waiting for
        //  another lane's through LDS, and takes its row of [A B] from two packed words set once: the table look-ups by lane index
        //  were vector loads from constant memory in every stage)
        constexpr int FWDU = NC > 0 ? MMPC_GEN_FWD_UNROLL : 1;
#pragma unroll FWDU
        for (int k = 0; k < N; k++) {
            LANES_BEGIN
            if (lane < NX) {
                const unsigned st = MMPC_LS.fw_st, fi = MMPC_LS.fw_in;
                const int a = (int)(fi & 255u) - 1;
                const double *cv = CV + k * MMPC_NCV, *dx = DX + k * NX;
                double v = CD[k * NX + lane];
                for (int q = 0; q < 4; q++) v += cv[(st >> (8 * q + 4)) & 15u] * dx[(st >> (8 * q)) & 15u];
                if (a >= 0) {
                    double du = KF[k * NU + a];
                    for (int j = 0; j < NX; j++) du += KK[(k * NU + a) * NX + j] * dx[j];
                    v += cv[(fi >> 8) & 255u] * du;
                    if ((fi >> 16) & 1u) DU[k * NU + a] = du;
                }
                DX[(k + 1) * NX + lane] = v;
            }
            LANES_END
        }
#else
        for (int k = 0; k < N; k++) {
            LANES_BEGIN
            if (lane < NU) {
                double v = KF[k * NU + lane];
                for (int j = 0; j < NX; j++) v += KK[(k * NU + lane) * NX + j] * DX[k * NX + j];
                DU[k * NU + lane] = v;
            }
            LANES_END
            LANES_BEGIN
            if (lane < NX) {
                const double *cv = CV + k * MMPC_NCV;
                double v = CD[k * NX + lane];
                for (int q = 0; q < 5; q++) {
                    const int c = TB::rcol(lane, q);
                    v += cv[TB::rcv(lane, q)] * (c < NX ? DX[k * NX + c] : DU[k * NU + c - NX]);
                }
                DX[(k + 1) * NX + lane] = v;
            }
            LANES_END
        }
#endif
        MMPC_GS(4)
        // ---- D1: multiplier step and slack-variable step
        LANES_BEGIN
        for (int k = lane; k < NS; k += MMPC_WAVE) {
            const double *dx = DX + k * NX;
            double vdx = 0.0;
            const int ny = NSELF ? 6 : 2;
            for (int a = 0; a < ny; a++) vdx += VX[k * 6 + a] * dx[kY[a]];
            if (k == N - 1 && NSELF) for (int a = 0; a < 6; a++) vdx += VXN[a] * DX[N * NX + kY[a]];
            bool back_self = false;
            if (NQ && k >= 1) {
                for (int a = 0; a < 6; a++) vdx += VQ[k * 6 + a] * DX[(k - 1) * NX + kY[a]];
                if (k != N - 1) for (int a = 0; a < 6; a++) back_self = back_self || VQ[k * 6 + a] != 0.0;
            }
            const double dsk = -(GSS[k] - vdx) / HSS[k];
            DS[k] = dsk;
            for (int i = 0; i < NX; i++) {
                double v = QX[k * NX + i];
                for (int j = 0; j < NX; j++)
                    v += HXX[k * NXX + (i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i)] * dx[j];
                if (teq) v += PNUS[(k * NX + i) * MMPC_NBC] * NUEQ[2] + PNUS[(k * NX + i) * MMPC_NBC + 1] * NUEQ[3];
                if (NQ && SIGW[0] != 0.0) v += PNUS[(k * NX + i) * MMPC_NBC + 2] * SIGW[1];
                DLAM[k * NX + i] = -v - LAM[k * NX + i];
            }
            // a slack eliminated one stage earlier is not part of this stage's cost-to-go: its pull on x_k enters the multiplier of
            // the dynamics directly,  lam_k+ = -(P_k dx_k + p_k) + v_k ds_k
            if (back_self) for (int a = 0; a < 6; a++) DLAM[k * NX + kY[a]] += VX[k * 6 + a] * dsk;
        }
        LANES_END
        MMPC_GS(5)
        // ---- D2: row steps, fraction-to-boundary, directional derivative, merit at alpha = 0
        const double tau = mmpc_max(0.99, 1.0 - mu);
        auto stage_merit = [&](int k, double alpha, double &phi_k, double &th_k, int soc_acc = 0, double a_prev = 0.0) {
            // barrier objective and l1 infeasibility contributions of stage k at w + alpha dw
            double xk[NX], uk[NU > 0 ? NU : 1];
            for (int j = 0; j < NX; j++) xk[j] = X[k * NX + j] + alpha * DX[k * NX + j];
            const double sk = S[k] + alpha * DS[k];
            const int ks = slack_idx(k);
            const double sks = S[ks] + alpha * DS[ks];
            double f = Sw * sk * sk, th = 0.0;
            f += 0.5 * mmpc_state_cost<KIND>(WTS, k == N, xk, XREF + k * NREF, nullptr, nullptr, false);
            double sn, cs;
            MMPC_SINCOS(xk[2], &sn, &cs);
            if (k < N) {
                for (int a = 0; a < NU; a++) uk[a] = U[k * NU + a] + alpha * DU[k * NU + a];
                double q = 0.0;
                for (int a = 0; a < NU; a++) for (int b = 0; b < NU; b++)
                    q += (uk[a] - UREF[k * NU + a]) * WTS[MMPC_W_R2 + a * NU + b] * (uk[b] - UREF[k * NU + b])
                       + (uk[a] - ULAST[k * NU + a]) * WTS[MMPC_W_W2 + a * NU + b] * (uk[b] - ULAST[k * NU + b]);
                f += 0.5 * q;
                double xn[NX];
                xn[0] = xk[0] + dt * xk[3]; xn[1] = xk[1] + dt * xk[4]; xn[2] = xk[2] + dt * xk[5];
                xn[3] = xk[3] + dt * (uk[0] * cs - xk[4] * xk[5]);
                xn[4] = xk[4] + dt * (uk[0] * sn + xk[3] * xk[5]);
                xn[5] = xk[5] + dt * uk[1];
                if (KIND != 1) { xn[6] = xk[6] + dt * uk[2]; xn[7] = xk[7] + dt * uk[3]; xn[8] = xk[8] + dt * uk[4]; }
                for (int j = 0; j < NX; j++) {
                    const double cj = xn[j] - (X[(k + 1) * NX + j] + alpha * DX[(k + 1) * NX + j]);
                    th += fabs(cj);
                    if (soc_acc) CD[k * NX + j] = a_prev * CD[k * NX + j] + cj;   // c_soc <- a c_soc + c(trial)
                }
            }
            // sum of log t: product of mantissas + sum of exponents (one log per stage)
            double mant = 1.0; int ex = 0;
            auto acc = [&](double tv) { int e2; mant *= frexp(tv, &e2); ex += e2; if (mant < 1e-200) { mant = frexp(mant, &e2); ex += e2; } };
            for (int r = 0; r < SL_C; r++) {
                double b;
                if (!box_bound(k, r, b)) continue;
                double h, dv;
                if (r < NU) { h = b - uk[r]; dv = -DU[k * NU + r]; }
                else if (r < SL_XLO) { h = uk[r - NU] - b; dv = DU[k * NU + r - NU]; }
                else if (r < SL_XHI) { h = b - xk[r - SL_XLO]; dv = -DX[k * NX + r - SL_XLO]; }
                else { h = xk[r - SL_XHI] - b; dv = DX[k * NX + r - SL_XHI]; }
                // at alpha: t + alpha dt, with dt = -(h0 + t) - dv and h linear:  h(alpha) = h0 + alpha dv
                const double t0 = T[k * R + r];
                const double h0 = h - alpha * dv;
                const double tv = t0 + alpha * (-(h0 + t0) - dv);
                th += fabs(h + tv);
                acc(tv);
            }
            double hr[HRMAX];
            {
                for (int m = 0; m < M; m++) {
                    const double *o = obs_ptr(k, m);
                    const double dx = xk[0] - o[0], dy = xk[1] - o[1];
                    hr[m] = (o[2] + MMPC_BASE_R) - sqrt(dx * dx + dy * dy) - sk;
                }
                if (NSELF) {
                    double dr[3], dz[3];
                    mmpc_arm_segments(xk[NX - 3], xk[NX - 2], xk[NX - 1], dr, dz);
                    for (int i = 0; i < NSELF; i++) hr[M + i] = mmpc_self_row(i, xk[0], xk[1], cs, sn, dr, dz, nullptr) - sks;
                    for (int i = 0; i < NHS; i++) hr[M + NSELF + i] = mmpc_hs_row(P, i, xk[0], xk[1], cs, sn, dr, dz, nullptr) - sk;
                    if (NQ && k >= 1) {
                        double xp[NX], drp[3], dzp[3], snp, csp;
                        for (int j = 0; j < NX; j++) xp[j] = X[(k - 1) * NX + j] + alpha * DX[(k - 1) * NX + j];
                        MMPC_SINCOS(xp[2], &snp, &csp);
                        mmpc_arm_segments(xp[NX - 3], xp[NX - 2], xp[NX - 1], drp, dzp);
                        for (int e = 0; e < NQ; e++) { int br; hr[M + NSELF + NHS + e] = q8_row(e, xk, cs, sn, dr, dz, xp, csp, snp, drp, dzp, br, nullptr, nullptr) - sk; }
                    }
                }
            }
            for (int m = 0; m < NR; m++) {
                if (k == 0 && m >= NR - NQ) continue;   // (no rows of the as-written set at stage 0)
                const double t0 = T[k * R + SL_C + m], tv = t0 + alpha * DTR[k * NR + m];
                th += fabs(hr[m] + tv);
                acc(tv);
                if (soc_acc) HR[k * NR + m] = a_prev * (HR[k * NR + m] + t0) + (hr[m] + tv) - t0;   // (h + t)_soc, kept as h_soc = . - t
            }
            if (teq && k == N) for (int j = 0; j < 2; j++) th += fabs(xk[j] - XREF[N * NX + j]);
            phi_k = f - mu * (log(mant) + (double)ex * 0.69314718055994530942);
            th_k = th;
        };
        LANES_BEGIN
        double ap = 1.0, ad = 1.0, dphi = 0.0;
        // one item per (row, stage), row-major: the lanes of a trip run the same row type on different stages (three rows of NS
        // stages per trip) instead of 21 lanes walking through all the rows of their stage
        for (int item = lane; item < R * NS; item += MMPC_WAVE) {
            const int r = item / NS, k = item - r * NS;
            const double *dx = DX + k * NX;
            {
                double h, jd, b;
                if (r < SL_C) {
                    if (!box_bound(k, r, b)) continue;
                    if (r < NU) { h = b - U[k * NU + r]; jd = -DU[k * NU + r]; }
                    else if (r < SL_XLO) { h = U[k * NU + r - NU] - b; jd = DU[k * NU + r - NU]; }
                    else if (r < SL_XHI) { h = b - X[k * NX + r - SL_XLO]; jd = -dx[r - SL_XLO]; }
                    else { h = X[k * NX + r - SL_XHI] - b; jd = dx[r - SL_XHI]; }
                } else if (r < SL_S) {
                    const int m = r - SL_C;
                    h = HR[k * NR + m];
                    jd = GC[(k * M + m) * 2] * dx[0] + GC[(k * M + m) * 2 + 1] * dx[1] - DS[k];
                } else if (r < SL_H) {
                    const int i = r - SL_S;
                    h = HR[k * NR + M + i];
                    jd = -DS[slack_idx(k)];
                    for (int a = 0; a < 6; a++) jd += GSF[(k * NSELF + i) * 6 + a] * dx[kY[a]];
                } else if (r < SL_Q) {
                    const int i = r - SL_H;
                    h = HR[k * NR + M + NSELF + i];
                    jd = -DS[k];
                    for (int a = 0; a < 6; a++) jd += GHS[(k * 6 + i) * 6 + a] * dx[kY[a]];
                } else {
                    if (k == 0) { DTR[k * NR + r - SL_C] = 0.0; continue; }
                    const int e = r - SL_Q;
                    const double *dq = DX + (BQ8[k * NQ + e] != 0.0 ? k - 1 : k) * NX;
                    h = HR[k * NR + M + NSELF + NHS + e];
                    jd = -DS[k];
                    for (int a = 0; a < 6; a++) jd += GQ8[(k * NQ + e) * 6 + a] * dq[kY[a]];
                }
                // (reciprocals instead of IEEE divisions, as in the specialised kernel: the five quotients of a row were most of this phase)
                const double t = T[k * R + r], z = Z[k * R + r], it_ = mmpc_rcp(t);
                const double dtv = -(h + t) - jd, dzv = mu * it_ - z - (z * it_) * dtv;
                if (r >= SL_C) DTR[k * NR + r - SL_C] = dtv;
                if (dtv < 0) ap = mmpc_min(ap, -tau * t * mmpc_rcp(dtv));
                if (dzv < 0) ad = mmpc_min(ad, -tau * z * mmpc_rcp(dzv));
                dphi -= mu * dtv * it_;
            }
        }
        for (int k = lane; k < NS; k += MMPC_WAVE) {
            const double *dx = DX + k * NX;
            for (int j = 0; j < NX; j++) dphi += GX[k * NX + j] * dx[j];
            if (k < N) for (int a = 0; a < NU; a++) dphi += GU[k * NU + a] * DU[k * NU + a];
            dphi += 2 * Sw * S[k] * DS[k];
        }
        RED[0 * MMPC_WAVE + lane] = ap; RED[1 * MMPC_WAVE + lane] = ad; RED[2 * MMPC_WAVE + lane] = dphi;
        LANES_END
        double ap = 1.0, ad = 1.0, dphi = 0.0;
        ap = MMPC_GRED_MIN(RED + 0 * MMPC_WAVE); ad = MMPC_GRED_MIN(RED + 1 * MMPC_WAVE); dphi = MMPC_GRED_SUM(RED + 2 * MMPC_WAVE);
        MMPC_GS(6)
        if (soc_p == 0) {
            ap0 = ap; ad0 = ad; dphi0 = dphi;
            // ---- merit at the current point: from the evaluation E1 (same point, same slacks), with the barrier parameter as it is now
            phi0 = m_F0 - mu * m_LG0; th0 = m_TH0;
            if (!mmpc_finite(phi0) || !mmpc_finite(th0)) { status = 2; fatal = true; break; }   // (an infinite reference / obstacle: opti.solve() raises)
            if (!filt_init) { nfilt = 0; th_max = 1e4 * mmpc_max(1.0, th0); th_min = 1e-4 * mmpc_max(1.0, th0); filt_init = 1; }
        }
        MMPC_GS(7)
        // ---- filter line search (Waechter-Biegler acceptance rules, second-order correction, no restoration phase)
        // barrier objective and infeasibility of the trial point at step a along the direction in DX .. DTR (soc_acc: also
        // c_soc <- a_prev c_soc + c(trial) into CD / HR)
        auto trial = [&](double a, double &phi, double &th, int soc_acc, double a_prev) {
            LANES_BEGIN
            double ph = 0.0, tq = 0.0;
            for (int k = lane; k < NS; k += MMPC_WAVE) { double u, v; stage_merit(k, a, u, v, soc_acc, a_prev); ph += u; tq += v; }
            RED[5 * MMPC_WAVE + lane] = ph; RED[6 * MMPC_WAVE + lane] = tq;
            LANES_END
            phi = MMPC_GRED_SUM(RED + 5 * MMPC_WAVE); th = MMPC_GRED_SUM(RED + 6 * MMPC_WAVE);
        };
        // acceptance to the filter and against the current point; a0: the step length of the switching and Armijo conditions (a
        // corrected step is tested with the length of the step it corrects)
        auto accept = [&](double th, double phi, double a0) -> bool {
            bool okf = th < th_max, accepted = false;
            for (int i = 0; i < nfilt && okf; i++) if (th >= FILT[2 * i] && phi >= FILT[2 * i + 1]) okf = false;
            const bool ftype = dphi0 < 0 && th0 <= th_min && a0 * pow(-dphi0, 2.3) > pow(th0, 1.1);
            bool augment = false;
            if (okf) {
                if (ftype) accepted = phi <= phi0 + 1e-8 * a0 * dphi0 + 1e-14 * fabs(phi0);
                else if (th <= (1 - 1e-5) * th0 || phi <= phi0 - 1e-5 * th0) { accepted = true; augment = true; }
            }
            if (augment) {
                int slot = nfilt;
                if (nfilt >= MMPC_FCAP) { slot = 0; for (int i = 1; i < MMPC_FCAP; i++) if (FILT[2 * i] > FILT[2 * slot]) slot = i; }
                else nfilt++;
                LANES_BEGIN
                if (lane == 0) { FILT[2 * slot] = (1 - 1e-5) * th0; FILT[2 * slot + 1] = phi0 - 1e-5 * th0; }
                LANES_END
            }
            return accepted;
        };
        // backtracking along the uncorrected direction from trial ls0 of the first pass on (+ the filter-reset heuristic)
        auto backtrack = [&](int ls0) {
            for (int lspass = 0; lspass < 2; lspass++) {
                bool accepted = false;
                const int lsb = lspass == 0 ? ls0 : 0;
                alpha = ap0;
                for (int q = 0; q < lsb; q++) alpha *= 0.5;
                for (int ls = lsb; ls < MMPC_MAX_LS; ls++) {
                    double phi, th;
                    trial(alpha, phi, th, 0, 0.0);
                    accepted = accept(th, phi, alpha);
                    if (accepted) break;
                    if (ls < MMPC_MAX_LS - 1) alpha *= 0.5;
                }
                if (accepted || nfilt == 0) break;
                nfilt = 0;  // filter reset heuristic: the filter blocked every trial step
            }
        };
        // what the correction keeps of the uncorrected direction (restored when it fails)
        auto soc_swap = [&](bool save) {
            double *q = io.soc;
            LANES_BEGIN
            const int o1 = NS * NX, o2 = o1 + NS * NU, o3 = o2 + NS, o4 = o3 + NS * NX, o5 = o4 + NS * NR;
            for (int i = lane; i < NS * NX; i += MMPC_WAVE) { if (save) { q[i] = DX[i]; q[o3 + i] = DLAM[i]; } else { DX[i] = q[i]; DLAM[i] = q[o3 + i]; } }
            for (int i = lane; i < N * NU; i += MMPC_WAVE) { if (save) q[o1 + i] = DU[i]; else DU[i] = q[o1 + i]; }
            for (int i = lane; i < NS; i += MMPC_WAVE) { if (save) q[o2 + i] = DS[i]; else DS[i] = q[o2 + i]; }
            for (int i = lane; i < NS * NR; i += MMPC_WAVE) { if (save) q[o4 + i] = DTR[i]; else DTR[i] = q[o4 + i]; }
            if (lane == 0) {
                if (save) { q[o5] = NUEQ[2]; q[o5 + 1] = NUEQ[3]; q[o5 + 2] = SIGW[1]; }
                else { NUEQ[2] = q[o5]; NUEQ[3] = q[o5 + 1]; SIGW[1] = q[o5 + 2]; }
            }
            LANES_END
        };
        if (soc_p == 0) {
            double phi, th;
            alpha = ap;
            trial(alpha, phi, th, 0, 0.0);
            if (accept(th, phi, alpha)) break;
            if (MMPC_SOC_MAX > 0 && !teq && io.soc && th >= th0 && th0 <= th_min) {
                row_step(alpha, ad0, false, true);    // the multipliers' step, safeguarded with the trial slacks
                z_done = true;
                soc_swap(true);
                trial(alpha, phi, th, 1, alpha);      // CD, HR <- alpha c(x_k) + c(x_k + alpha d)
                th_prev = th; soc_p = 1;
                continue;
            }
            backtrack(1);
            break;
        } else {
            double phi, th;
            trial(ap, phi, th, 0, 0.0);
            if (accept(th, phi, ap0)) { alpha = ap; break; }
            if (th > 0.99 * th_prev || soc_p >= MMPC_SOC_MAX) {   // the correction failed: on along the uncorrected direction
                soc_swap(false);
                backtrack(1);
                break;
            }
            th_prev = th;
            trial(ap, phi, th, 1, ap);
            soc_p++;
        }
        }   // direction passes
        if (fatal) break;
#ifdef MMPC_EMU_DEBUG
        fprintf(stderr, "generic it %d mu %.2e E0 %.3e err_d %.3e err_p %.3e alpha %.3e ap %.3e ad %.3e prox %.1e dphi %.3e soc %d\n", it, mu, E0, err_d, err_p, alpha, ap0, ad0, prox, dphi0, soc_p);
#endif
        if (!teq) mmpc_prox_update(ap0, alpha, prox, nsmall);   // (with the terminal equality the forced correction makes nu grow like prox)
        MMPC_GS(8)
        // ---- update (one item per (row, stage), row-major as in D2)
        row_step(alpha, ad0, true, !z_done);
        LANES_BEGIN
        for (int i = lane; i < NS * NX; i += MMPC_WAVE)
            if (i >= NX) { X[i] += alpha * DX[i]; LAM[i] += alpha * DLAM[i]; }
        for (int i = lane; i < N * NU; i += MMPC_WAVE) U[i] += alpha * DU[i];
        for (int i = lane; i < NS; i += MMPC_WAVE) S[i] += alpha * DS[i];
        if (teq && lane < 2) NUEQ[lane] += alpha * (NUEQ[2 + lane] - NUEQ[lane]);
        LANES_END
    }

    MMPC_GS(9) MMPC_GEND()
    // ---------------------------------------------------------------- results
    LANES_BEGIN
    double f = 0.0;
    for (int k = lane; k < NS; k += MMPC_WAVE) {
        const double *xk = X + k * NX;
        double q = mmpc_state_cost<KIND>(WTS, k == N, xk, XREF + k * NREF, nullptr, nullptr, false);
        if (k < N) {
            const double *uk = U + k * NU;
            for (int a = 0; a < NU; a++) for (int b = 0; b < NU; b++)
                q += (uk[a] - UREF[k * NU + a]) * WTS[MMPC_W_R2 + a * NU + b] * (uk[b] - UREF[k * NU + b])
                   + (uk[a] - ULAST[k * NU + a]) * WTS[MMPC_W_W2 + a * NU + b] * (uk[b] - ULAST[k * NU + b]);
        }
        f += 0.5 * q + Sw * S[k] * S[k];
    }
    RED[lane] = f;
    for (int i = lane; i < NS * NX; i += MMPC_WAVE) io.X[i] = X[i];
    for (int i = lane; i < N * NU; i += MMPC_WAVE) io.U[i] = U[i];
    for (int i = lane; i < NS; i += MMPC_WAVE) io.s[i] = S[i];
    LANES_END
    double cost = 0.0;
    cost = MMPC_GRED_SUM(RED);
    LANES_BEGIN
    if (lane == 0) { *io.status = status; *io.iters = it; *io.cost = cost; *io.err = E0; }
    LANES_END
}
