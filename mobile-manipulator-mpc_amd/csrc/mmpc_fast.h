// mmpc_fast.h - performance path of the one-wavefront-per-problem interior-point MPC solver.
//
// Same algorithm, same results as mmpc_core.h (the generic path; see that file for the
// reference citations), restructured for gfx950:
//   * horizon N and obstacle count are template parameters; (x_k,u_k) are stored interleaved ("XU", NV = nx+nu
//     doubles per stage) so that a (stage, variable) pair is one linear index;
//   * the state of every inequality row lives in REGISTERS:
//       - box rows: lane <-> (stage, variable) pair, NPASS pairs per lane: the two multipliers and the two bounds (the
//         slack is the distance to the bound - the initial point is pushed inside, see MMPC_BOUND_PUSH);
//       - circle / self-collision rows: lane <-> stage: slack, multiplier, slack step;
//     LDS keeps only what is exchanged between lanes (trajectory, stage Hessians -> cost-to-go
//     matrices, feedback gains, search direction) : ~39 KB for N=20 -> 4 problems per CU;
//   * per-launch constants are copied into LDS once (no global/constant loads in the loop);
//   * the Riccati recursion runs on v_mfma_f64_16x16x4_f64 tiles with the cost-to-go matrix kept in accumulator layout
//     (see MMPC_MFMA below); wave-wide reductions use cross-lane shuffles;
//   * a line-search trial point is evaluated once, fully: the evaluation of the accepted trial is the evaluation the
//     next iteration starts from.
// Written in the same PHASE discipline as mmpc_core.h so that -DMMPC_EMU builds run it on the
// host (tests only); lane-private state that survives a phase lives in `LS` (struct per lane)
// and `WR` (reduction inputs).
#pragma once
#include "mmpc_core.h"

// Diagnostic build only (-DMMPC_STAMP): per-phase wave-cycle accounting, accumulated in registers and added to a
// __device__ array at the end (read by tools/probe_stamps.py).  The shipped kernel contains no stamps.
#if defined(MMPC_STAMP) && !defined(MMPC_EMU)
__device__ unsigned long long mmpc_stamp_acc[16];
#define MMPC_T0() unsigned long long t_prev_ = __builtin_readcyclecounter(), t_now_, t_acc_[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define MMPC_TS(i) { t_now_ = __builtin_readcyclecounter(); t_acc_[i] += t_now_ - t_prev_; t_prev_ = t_now_; }
#define MMPC_TEND() { if (threadIdx.x == 0) for (int i_ = 0; i_ < 14; i_++) atomicAdd(&mmpc_stamp_acc[i_], t_acc_[i_]); }
#else
#define MMPC_T0()
#define MMPC_TS(i)
#define MMPC_TEND()
#endif
// A wave's stores to global memory (the gain block of long horizons) followed by loads of the same words from other lanes of the
// SAME wave: the vector memory instructions of a wavefront are issued and processed in order, so the wavefront-scope fence of
// LANES_END (compiler ordering only, no s_waitcnt) is all that is needed - waiting for the stores' acknowledgement here cost 2 x
// ~1 k cycles per iteration (-DMMPC_GFENCE_WG restores the workgroup-scope fence for A/B checks).
#if defined(MMPC_GFENCE_WG) && !defined(MMPC_EMU)
#define MMPC_GFENCE() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#else
#define MMPC_GFENCE()
#endif
#define MMPC_MC_MAX 8    // largest number of circle rows the register-resident path is instantiated for

// (mmpc_sincos: mmpc_core.h)
// max / min of this path: v_max_f64 / v_min_f64 on the device (the ternary form of mmpc_core.h costs a compare and two selects
// per use - it hands a NaN in its second argument through; here NaNs are caught by the sums of the evaluation instead, see the
// status 2 test of the main loop)
#ifdef MMPC_EMU
MMPC_DEV double mmpc_vmax(double a, double b) { return a > b ? a : b; }
MMPC_DEV double mmpc_vmin(double a, double b) { return a < b ? a : b; }
#else
MMPC_DEV double mmpc_vmax(double a, double b) { return __builtin_fmax(a, b); }
MMPC_DEV double mmpc_vmin(double a, double b) { return __builtin_fmin(a, b); }
#endif
// multiplier safeguard (mmpc_z_safeguard) as a clamp of z to [mu / (kappa t), kappa mu / t] with one reciprocal instead of
// two IEEE divisions (38 inlined call sites in the trial-point phase)
MMPC_DEV double mmpc_z_safeguard_fast(double z, double t, double mu) {
    const double rt = mmpc_rcp(t);
    return mmpc_vmin(mmpc_vmax(z, (mu * (1.0 / MMPC_KAPPA_SIGMA)) * rt), (MMPC_KAPPA_SIGMA * mu) * rt);
}
// natural log of a product of mantissas m in (0,1]: renormalise to [sqrt(1/2), sqrt(2)), then
// log m = 2 atanh(s), s = (m-1)/(m+1), |s| < 0.1716, odd series to s^21 (< 1e-17)
MMPC_DEV double mmpc_log_mant(double m, int *ex) {
    int e;
    m = frexp(m, &e);
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    *ex += e;
    const double s = (m - 1.0) * mmpc_rcp(m + 1.0), z = s * s;
    double p = 1.0 / 21.0;
    p = fma(p, z, 1.0 / 19.0); p = fma(p, z, 1.0 / 17.0); p = fma(p, z, 1.0 / 15.0); p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0); p = fma(p, z, 1.0 / 9.0); p = fma(p, z, 1.0 / 7.0); p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0); p = fma(p, z, 1.0);
    return 2.0 * s * p;
}
// planar arm segments (as mmpc_arm_segments in mmpc_core.h) with the light-weight sincos
MMPC_DEV void mmpc_arm_segments_fast(double q1, double q2, double q3, double dr[3], double dz[3]) {
    double s1, c1, sA, cA, sB, cB;
    mmpc_sincos(q1, &s1, &c1);
    mmpc_sincos(q1 - q2, &sA, &cA);
    mmpc_sincos(q1 - q2 - q3, &sB, &cB);
    dr[0] = MMPC_A2 * s1 + MMPC_A3 * c1;
    dz[0] = MMPC_A2 * c1 - MMPC_A3 * s1;
    dr[1] = -MMPC_A3 * cA + MMPC_A5 * sA;
    dz[1] = MMPC_A3 * sA + MMPC_A5 * cA;
    dr[2] = MMPC_A6 * cB - MMPC_A7 * sB;
    dz[2] = -MMPC_A6 * sB - MMPC_A7 * cB;
}
// state entries the forward kinematics depends on (x, y, psi, q1, q2, q3), as a constant expression
MMPC_HD constexpr int mmpc_y(int a) { return a < 3 ? a : a + 3; }

#ifndef MMPC_UNROLL_NMAX
#define MMPC_UNROLL_NMAX 30  // (round 2, after the Riccati rewrite: the N = 30 instantiation gains 10.5 % from the unrolled loops; round 1: it lost)
#endif
#ifndef MMPC_FWD_UNROLL
#define MMPC_FWD_UNROLL 64  // stages per trip of the forward roll-out loop; >= N: fully unrolled (measured at N = 20: 2, 4, 10, 20 -> 1098, 1108,
                            // 1104, 1124 k solves/s; at N = 30 on C5: 1, 2, 5, 10, 30 -> 197, 214, 214, 215, 218 k solves/s)
#endif
#ifndef MMPC_RIC_UNROLL
#define MMPC_RIC_UNROLL 2   // stages per trip of the Riccati loop (measured best of 1, 2, 4, 5)
#endif
#define MMPC_B(x, q) (((x) >> (8 * (q))) & 255u)
// slack of a box row = distance of the variable to its bound.  The fraction-to-boundary rule keeps it positive in exact
// arithmetic ((1-tau) t with 1-tau down to 1e-9), but v + alpha dv is rounded to the grid of v (4e-16 near |v| = 2), so a
// distance of that size can round to zero: floor it at a couple of ulps.
MMPC_DEV double mmpc_box_t(double d) { return mmpc_vmax(d, 1e-15); }

#ifndef MMPC_SLIM_NMIN
#define MMPC_SLIM_NMIN 21  // horizons from here on: "slim" LDS layout (see mmpc_fast_layout) and circle rows spread over lanes
#endif
template <int KIND, int N>
struct MmpcFastDims {
    typedef MmpcDims<KIND> D;
    static constexpr int NX = D::NX, NU = D::NU, NV = D::NV, NXX = D::NXX, NUU = D::NUU, NSELF = D::NSELF;
    static constexpr int NS = N + 1;
    static constexpr int NPAIR = NS * NV;
    static constexpr int NPASS = (NPAIR + MMPC_WAVE - 1) / MMPC_WAVE;
    // Riccati recursion on 16x16 MFMA tiles in homogeneous form: tile index t < NX is state t, t = NX the constant 1 (its
    // row and column carry the gradient), NX < t <= NV input t-NX-1.  K-blocks (4 rows each) that cover the (x, 1) rows /
    // the input rows, accumulator registers that hold the input rows NX+1..NV of the stage matrix
    static constexpr int NKB = (NX + 1 + 3) / 4;
    static constexpr int NPU = NU * (NU - 1) / 2;   // couplings between the inputs of a stage kept for the gain back-substitution
    // input elimination legs of a Riccati stage: inputs whose tile rows (2q, 2q+1) share an accumulator register go in pairs
    static constexpr bool PAIRS = ((NX + 1) & 1) == 0;
    static constexpr int NLEG = PAIRS ? (NU + 1) / 2 : NU;
    // Long horizons and the base kind spread the circle rows of a stage over RG lanes (lane s NS + k owns the rows m = s + RG r
    // of stage k): those instantiations are short of registers (8 rows per stage; the 256-register cap of two waves per SIMD),
    // not of lanes.  (At most 3: the partial sums of the other
    // lanes travel through the first words of the stage's own - at that point dead - Hessian block.)  Short horizons keep the rows
    // in the stage lane, where their arithmetic fills the latency of the trigonometry around it (measured: -0.9 % at N = 20
    // when split, +11 % at N = 30, +2.6 % for the base kind).
    static constexpr int RG = ((N >= MMPC_SLIM_NMIN || KIND == 1) && 2 * NS <= MMPC_WAVE) ? (MMPC_WAVE / NS < 3 ? MMPC_WAVE / NS : 3) : 1;
    static_assert(NV + 1 <= 16, "stage matrix over (x, 1, u) must fit one 16x16 tile");
};


struct MmpcFastLayout {
    int XU, S, LAM, XUREF, ULAST, OBS, CST, CV, CD, TRG, HXX, QXU, HUXL, HUUL, HUX02, HUUD, SN, KK, KF, KU, DXU, DS, DLAM,
        DUMP, RB, RDS, Q1V, FILT, MISC, total;
};
// constants block (CST) offsets
#define MMPC_C_XLIM 0      // [2][9]
#define MMPC_C_ULIM 18     // [2][5]
#define MMPC_C_DULIM 28    // [2][5]
#define MMPC_C_WQ 38       // diag(Q2)[9]
#define MMPC_C_WP 47       // diag(P2)[9]
#define MMPC_C_WR 56       // diag(R2)[5]
#define MMPC_C_WW 61       // diag(W2)[5]
#define MMPC_C_RW2 66      // RW2 [25] (full)
#define MMPC_C_SIZE 92

// Long horizons (N >= MMPC_SLIM_NMIN) leave the read-only inputs that are touched once or twice per iteration - the
// reference trajectory, the previous inputs and the per-stage obstacle table - in HBM/L2 instead of LDS: at N = 30, M = 8 that is 63.2 -> 52.6 KB
// per problem, i.e. three resident problems per CU instead of two.
// Long horizons also keep the feedback gains (K_k, kf_k and the couplings between the inputs of a stage: 1 800 doubles at N = 30) in a
// per-instance block of global memory (L2 / MALL resident): the Riccati legs only store them, the back-substitution is one
// batched load / store pass and the roll-out fetches its gain row ahead of the chain.  At N = 30, M = 8 that is 52.6 -> 40.6 KB
// of LDS per problem: four resident problems per CU - one per SIMD - instead of three.
#ifndef MMPC_GK_MASK
#define MMPC_GK_MASK 0   // gain block: 1 = lanes without an entry of a pivot row do not store, 0 = they store to a dump slot of the block as the
                         // LDS variant does (measured on C5: the masked store - exec handling on the chain of every leg - 300 k against 314 k solves/s,
                         // for 10 % less write traffic out of L2)
#endif
#ifndef MMPC_GAINS_GLOBAL_NMIN
#define MMPC_GAINS_GLOBAL_NMIN MMPC_SLIM_NMIN
#endif
// the block: [KK: N NU NX][KF: N NU][KU: N NPU][dump: one slot per lane]
template <int KIND, int N>
struct MmpcGainBlock {
    typedef MmpcFastDims<KIND, N> F;
    static constexpr bool ON = N >= MMPC_GAINS_GLOBAL_NMIN;
    static constexpr int KK = 0, KF = KK + N * F::NU * F::NX, KU = KF + N * F::NU, DUMP = KU + N * F::NPU, total = ON ? DUMP + MMPC_WAVE : 0;
};
template <int KIND, int N>
MMPC_HD constexpr MmpcFastLayout mmpc_fast_layout(int M, int obs_per_stage) {
    typedef MmpcFastDims<KIND, N> F;
    constexpr bool SLIM = N >= MMPC_SLIM_NMIN;
    constexpr bool GK = MmpcGainBlock<KIND, N>::ON;
    MmpcFastLayout L{};
    int o = 0;
#define MMPC_CARVE(name, n) L.name = o; o += (n); o = (o + 1) & ~1;
    MMPC_CARVE(XU, F::NS * F::NV) MMPC_CARVE(S, F::NS) MMPC_CARVE(LAM, F::NS * F::NX)
    MMPC_CARVE(XUREF, SLIM ? 0 : F::NS * F::NV) MMPC_CARVE(ULAST, SLIM ? 0 : F::NS * F::NU)
    MMPC_CARVE(OBS, obs_per_stage ? (SLIM ? 0 : F::NS * M * 3) : M * 3)
    MMPC_CARVE(CST, MMPC_C_SIZE) MMPC_CARVE(CV, F::NS * MMPC_NCV) MMPC_CARVE(CD, F::NS * F::NX) MMPC_CARVE(TRG, F::NS * 8)
    MMPC_CARVE(HXX, F::NS * F::NXX) MMPC_CARVE(QXU, F::NS * F::NV) MMPC_CARVE(HUXL, F::NU * F::NX)
    MMPC_CARVE(HUUL, F::NUU) MMPC_CARVE(HUX02, F::NS) MMPC_CARVE(HUUD, F::NS * F::NU) MMPC_CARVE(SN, 16)
    MMPC_CARVE(KK, GK ? 0 : N * F::NU * F::NX) MMPC_CARVE(KF, GK ? 0 : N * F::NU) MMPC_CARVE(DXU, F::NS * F::NV) MMPC_CARVE(DS, F::NS)
    MMPC_CARVE(DLAM, F::NS * F::NX) MMPC_CARVE(RB, F::NS * F::NV) MMPC_CARVE(RDS, F::NS) MMPC_CARVE(Q1V, F::NV + 2) MMPC_CARVE(FILT, 2 * MMPC_FCAP)
    // scratch of the backward pass lives, where it fits, in arrays that are dead while it runs: the couplings between the
    // inputs of a stage in the search direction (written by the forward roll-out afterwards), the dump slots in the
    // multiplier step (D1)
    if (GK || F::NS * F::NV >= N * F::NPU) L.KU = L.DXU; else { MMPC_CARVE(KU, N * F::NPU) }
    if (F::NS * F::NX >= MMPC_WAVE) L.DUMP = L.DLAM; else { MMPC_CARVE(DUMP, MMPC_WAVE) } MMPC_CARVE(MISC, 8)
#undef MMPC_CARVE
    L.total = o;
    return L;
}

// doubles of the per-instance save area of a suspended solve: trajectory, slacks, equality multipliers, filter, scalars,
// and per lane the multipliers of its box rows and the slack / multiplier of its circle and self-collision rows
#define MMPC_NSCAL 12
template <int KIND, int N>
MMPC_HD int mmpc_fast_state_doubles(int MC) {
    typedef MmpcFastDims<KIND, N> F;
    return F::NPAIR + F::NS + F::NS * F::NX + 2 * MMPC_FCAP + MMPC_NSCAL + MMPC_WAVE * (2 * F::NPASS + 2 * (MC > 0 ? MC : 1) + 8);
}

template <int KIND, int N, int MC>
struct MmpcLaneState {
    typedef MmpcFastDims<KIND, N> F;
    // box pairs (stage, variable): index idx = lane + 64 p
    // (only the multipliers: the initial point is pushed >= MMPC_BOUND_PUSH inside every finite bound and the box rows are
    //  linear, so the slack of a box row IS the distance to its bound, t = v - lo / hi - v, for the whole solve)
    double lo_z[F::NPASS], hi_z[F::NPASS];
    // the bounds themselves (constant during a solve: the merged input box uses U_last); -+1e300 marks an absent side
    double b_lo[F::NPASS], b_hi[F::NPASS];
    // circle rows m = s + RG r of stage k, lane = s NS + k  (RG = 1: the rows of stage `lane`)
    static constexpr int MCR = MC > 0 ? (MC + F::RG - 1) / F::RG : 1;   // circle rows per lane
    double ct[MCR], cz[MCR], cdt[MCR];
    double cp[F::RG > 1 ? 9 : 1];                    // RG > 1: partial sums of this lane's circle rows, from a row phase to the stage phase after it
    // self-collision rows of stage `lane`
    double st[4], sz[4], sdt[4];
    // s_k elimination data of stage `lane`
    double hss, gss, vx[6];
    // Riccati recursion on MFMA tiles over (x, 1, u) (lane = 16 g + j; accumulator register r <-> row g + 4 r, column j)
    unsigned ab_o[F::NKB];                           // [A c B; 0 1 0] operand rows 4r+g, column j: LDS offset | stage stride << 16
    unsigned h_o[4], h_l[4];                         // stage-matrix entry of register r: LDS offset | stage stride << 16; offset of its
                                                     // stage-(N-1) extra term (Q1 elimination of s_{N-1}) or of the constant 0
    unsigned h_m;                                    // bit r: register r is part of [P p; p^T .]; bit 4+r: ... and is stored (lower triangle, p)
    unsigned p_o[4];                                 // where register r of [P_k | p_k] is stored: h_o[r] for the stored entries, a dump slot otherwise
    int kl_b[F::NLEG], kl_s[F::NLEG];                // where this lane stores its entry of the normalised pivot row(s) of leg l: LDS byte
                                                     // offset at stage 0 and per-stage stride (gain row / kf / coupling; a dump slot otherwise)
    MmpcAcc rP, rT, rM;                              // cost-to-go [P p; p^T .], its product T with the dynamics, stage matrix M
    double rAB[F::NKB], opa, opb;                    // MFMA operands
    double nab[F::NKB], nhm[4];                      // next stage's dynamics rows and stage-matrix entries (loaded one stage ahead)
    // forward roll-out, row `lane` of [A B] (base.py:19-26): dx+[i] = dx[i] + sum_{j=2..5} C_j dx[j] + C_u du_a + c[i];
    // f_v: coefficient ids (into CV[k]) of C_2..C_5, f_x: id of C_u << 8 | (a + 1) << 16 (a: the input of this row, -1 none)
    unsigned f_v, f_x;
    double fw[MMPC_FW_SLOTS];                        // dx_k[lane] during the roll-out (exchanged by v_readlane, not LDS)
    int fw_a[5];                                     // where the lane stores dx_{k+1}[lane] and du_k[a]: LDS offsets at k = 0 and their stage stride
                                                     // (set once per roll-out: the lane id is opaque to the compiler in every phase)
};

#ifndef MMPC_EMU
#if MMPC_RED_DPP
MMPC_DEV double mmpc_op_add(double a, double b) { return a + b; }
MMPC_WAVE_RED(mmpc_wave_sum, mmpc_op_add)
MMPC_WAVE_RED(mmpc_wave_max, mmpc_vmax)
MMPC_WAVE_RED(mmpc_wave_min, mmpc_vmin)
#else
MMPC_DEV double mmpc_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
MMPC_DEV double mmpc_wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = mmpc_vmax(v, __shfl_xor(v, o));
    return v;
}
MMPC_DEV double mmpc_wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = mmpc_vmin(v, __shfl_xor(v, o));
    return v;
}
#endif
#define MMPC_RED_SUM(i) mmpc_wave_sum(wr_one[i])
#define MMPC_RED_MAX(i) mmpc_wave_max(wr_one[i])
#define MMPC_RED_MIN(i) mmpc_wave_min(wr_one[i])
#else
// host emulation: butterfly in the same pairing order as the device shuffles, so that the
// floating-point sums are bit-identical on every "lane"
static inline double mmpc_emu_red(double (*wr)[9], int i, int op) {
    double v[MMPC_WAVE], w[MMPC_WAVE];
    for (int l = 0; l < MMPC_WAVE; l++) v[l] = wr[l][i];
    for (int q = 0; q < 6; q++) {
        const int o = MMPC_RED_DPP ? (1 << q) : (32 >> q);
        for (int l = 0; l < MMPC_WAVE; l++) {
            // partner of the device butterfly: lane ^ 1, ^ 2, the mirror inside 8 / 16 lanes, lane ^ 16, ^ 32  (ds_bpermute form: lane ^ o)
            const int p = !MMPC_RED_DPP ? (l ^ o) : (o == 4 ? ((l & ~7) | (7 - (l & 7))) : (o == 8 ? ((l & ~15) | (15 - (l & 15))) : (l ^ o)));
            const double a = v[l], b = v[p];
            w[l] = op == 0 ? a + b : (op == 1 ? (a > b ? a : b) : (a < b ? a : b));
        }
        for (int l = 0; l < MMPC_WAVE; l++) v[l] = w[l];
    }
    return v[0];
}
#define MMPC_RED_SUM(i) mmpc_emu_red(wr_all, i, 0)
#define MMPC_RED_MAX(i) mmpc_emu_red(wr_all, i, 1)
#define MMPC_RED_MIN(i) mmpc_emu_red(wr_all, i, 2)
#endif

// compile-time switch handed to the generic lambdas of the assembly / row-step phases (corrected pass of the second-order correction or not)
template <bool B> struct MmpcTag { static constexpr bool value = B; };

// product-of-mantissas accumulator for sum(log t)
struct MmpcLogAcc {
    double mant; int ex;
    MMPC_DEV void init() { mant = 1.0; ex = 0; }
    // (a lane multiplies at most a few dozen mantissas in [0.5, 1) between init() and value(): no underflow, no renormalisation)
    MMPC_DEV void mul(double t) { int e; mant *= frexp(t, &e); ex += e; }
    MMPC_DEV double value() const { int e = ex; const double l = mmpc_log_mant(mant, &e); return l + (double)e * 0.69314718055994530942; }
};

// CONT: built with the iteration budget / continuation code (a separate instantiation: the extra live values cost the
// default kernel registers it does not have)
// OPS: the configuration's obs_per_stage when it is a constant of the instantiation (the device kernels: every LDS offset is
// then an immediate), -1: read from the parameter block
template <int KIND, int N, int MC, bool CONT = false, int OPS = -1>
MMPC_DEV void mmpc_solve_fast(const MmpcParams &P, const MmpcIO io, double *lds MMPC_EMU_ARG) {
    typedef MmpcFastDims<KIND, N> F;
    typedef MmpcTab<KIND> TB;
    const int ops = OPS >= 0 ? OPS : P.obs_per_stage;
    constexpr int NX = F::NX, NU = F::NU, NV = F::NV, NXX = F::NXX, NUU = F::NUU, NSELF = F::NSELF, NS = F::NS;
    constexpr int NPASS = F::NPASS, NPAIR = F::NPAIR, NKB = F::NKB, NPU = F::NPU;
    constexpr int M = MC;   // number of circle obstacles is a template parameter on this path
    constexpr int RG = F::RG, MCR = MmpcLaneState<KIND, N, MC>::MCR;   // circle rows: RG lanes per stage, MCR rows per lane
// lane -> (row group rs, stage rk) of the circle rows; row slot r of the lane is obstacle rs + RG r
#define MMPC_ROW_LANE const int rs = RG > 1 ? lane / NS : 0, rk = lane - rs * NS; const bool rlane = M > 0 && lane < RG * NS;
    // stages per trip of the Riccati / forward loops: unrolling saves the per-stage pointer bumps and register shuffles,
    // but costs registers - it only pays where the kernel does not spill (measured per instantiation)
#ifdef MMPC_UNROLL_BASE
    constexpr bool ROOMY = N <= MMPC_UNROLL_NMAX;
#else
    constexpr bool ROOMY = KIND == 0 && N <= MMPC_UNROLL_NMAX;
#endif
    constexpr bool SLIM = N >= MMPC_SLIM_NMIN;   // references and per-stage obstacles are read from HBM/L2 (see mmpc_fast_layout)
    constexpr int RIC_UNROLL = ROOMY ? MMPC_RIC_UNROLL : 1, FWD_UNROLL = ROOMY ? (MMPC_FWD_UNROLL < N ? MMPC_FWD_UNROLL : N) : 1;
    const MmpcFastLayout L = mmpc_fast_layout<KIND, N>(M, ops);
    double *XU = lds + L.XU, *S = lds + L.S, *LAM = lds + L.LAM, *XUREF = lds + L.XUREF, *ULAST = lds + L.ULAST,
           *OBS = lds + L.OBS, *CST = lds + L.CST, *CV = lds + L.CV, *CD = lds + L.CD, *TRG = lds + L.TRG, *HXX = lds + L.HXX,
           *QXU = lds + L.QXU, *HUXL = lds + L.HUXL, *HUUL = lds + L.HUUL, *HUX02 = lds + L.HUX02,
           *HUUD = lds + L.HUUD, *SN = lds + L.SN, *DXU = lds + L.DXU,
           *DS = lds + L.DS, *DLAM = lds + L.DLAM, *FILT = lds + L.FILT;
    // gains and input couplings: LDS, or this instance's block of global memory (MmpcGainBlock)
    typedef MmpcGainBlock<KIND, N> GB;
    constexpr bool GK = GB::ON;
    double *const KBASE = GK ? io.gscr : lds;     // what the store offsets of the elimination legs (kl_b) are relative to
    double *const KK = GK ? io.gscr + GB::KK : lds + L.KK, *const KF = GK ? io.gscr + GB::KF : lds + L.KF, *const KU = GK ? io.gscr + GB::KU : lds + L.KU;
    constexpr int O_KK = GK ? GB::KK : 0, O_KF = GK ? GB::KF : 0, O_KU = GK ? GB::KU : 0, O_KDUMP = GK ? GB::DUMP : 0;
    // element `idx` of an array of the gain block: uniform base + 32-bit byte offset (the scalar-base form of the global
    // load / store: one address register per access instead of a 64-bit pair built per lane)
    auto gk = [](double *base, int idx) -> double & {
        if constexpr (GK) return *(double *)((char *)base + (size_t)((unsigned)idx * 8u)); else return base[idx];
    };
    double *const RB = lds + L.RB, *const RDS = lds + L.RDS, *const Q1V = lds + L.Q1V;   // residual base r[k][v] and the s_k residual of the current point
    const double dt = P.dt, Sw = P.S, tol = P.tol;
#ifdef MMPC_EMU
    static thread_local MmpcLaneState<KIND, N, MC> ls_all[MMPC_WAVE];
    static thread_local double wr_all[MMPC_WAVE][9];
#else
    MmpcLaneState<KIND, N, MC> ls_one;
    double wr_one[9];
#endif
    const double *const OBSP = (SLIM && ops) ? io.obs : OBS;
    auto obs_ptr = [&](int k, int m) -> const double * { return OBSP + ((ops ? k * M : 0) + m) * 3; };
    auto ulast_at = [&](int k, int a) -> double {
        if (!SLIM) return ULAST[k * NU + a];
        return k < N ? io.u_last[k * NU + a] : 0.0;
    };
    auto ref_at = [&](int idx, int k, int v) -> double {
        if (!SLIM) return XUREF[idx];
        return v < NX ? io.traj_ref[k * NX + v] : (k < N ? io.u_ref[k * NU + v - NX] : 0.0);
    };
    auto slack_idx = [&](int k) -> int { return k < N - 1 ? k : N - 1; };

    // ------------------------------------------------------------------ load
    LANES_BEGIN
    for (int i = lane; i < NPAIR; i += MMPC_WAVE) {
        const int k = i / NV, v = i % NV;
        double val, ref;
        if (v < NX) {
            double x0 = io.x_init[v];
            if (KIND == 0) x0 = x0 > P.xlim[1][v] ? P.xlim[1][v] : (x0 < P.xlim[0][v] ? P.xlim[0][v] : x0);   // (a NaN stays a NaN: status 2)
            val = (io.x_guess && k >= 1) ? io.x_guess[k * NX + v] : x0;
            ref = io.traj_ref[k * NX + v];
        } else {
            val = k < N ? (io.u_guess ? io.u_guess[k * NU + v - NX] : io.u_last[k * NU + v - NX]) : 0.0;
            ref = k < N ? io.u_ref[k * NU + v - NX] : 0.0;
        }
        XU[i] = val;
        if (!SLIM) XUREF[i] = ref;
    }
    if (!SLIM)
        for (int i = lane; i < NS * NU; i += MMPC_WAVE) ULAST[i] = i < N * NU ? io.u_last[i] : 0.0;
    for (int i = lane; i < NS * NX; i += MMPC_WAVE) LAM[i] = 0.0;
    for (int i = lane; i < NS; i += MMPC_WAVE) S[i] = 0.0;
    if (!(SLIM && ops))
        for (int i = lane; i < (ops ? NS : 1) * M * 3; i += MMPC_WAVE) OBS[i] = io.obs[i];
    for (int i = lane; i < MMPC_C_SIZE; i += MMPC_WAVE) {
        double v = 0.0;
        if (i < 18) v = P.xlim[i / 9][i % 9];
        else if (i < 28) v = P.ulim[(i - 18) / 5][(i - 18) % 5];
        else if (i < 38) v = P.dulim[(i - 28) / 5][(i - 28) % 5];
        else if (i < 47) { const int j = i - 38; v = j < NX ? P.Q2[j * NX + j] : 0.0; }
        else if (i < 56) { const int j = i - 47; v = j < NX ? P.P2[j * NX + j] : 0.0; }
        else if (i < 61) { const int j = i - 56; v = j < NU ? P.R2[j * NU + j] : 0.0; }
        else if (i < 66) { const int j = i - 61; v = j < NU ? P.W2[j * NU + j] : 0.0; }
        else if (i < 66 + 25) v = P.RW2[i - 66];
        CST[i] = v;
    }
    // Riccati index tables (once)
    {
        auto &ls = MMPC_LS;
        const int g = lane >> 4, j = lane & 15;
        // tile index -> index in the (x, u | gradient) numbering of the stage arrays: state t, gradient NV, input NX+a, 99 = outside
        auto lg = [&](int t) -> int { return t < NX ? t : t == NX ? NV : t <= NV ? t - 1 : 99; };
        const int jl = lg(j);
#pragma unroll
        for (int r = 0; r < NKB; r++) {
            // operand entry of the homogeneous dynamics [A B c; 0 0 1] (rows over (x, 1), columns over (x, 1, u)), row m = 4r+g:
            // a coefficient of CV[k] (ids 0,1,2 are the constants 0,1,dt), the defect c_k[m] in the column of the 1, the 1
            // itself, or the constant 0 (stride 0) outside the matrix
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
        ls.h_m = 0u;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // stage matrix [Hxx Hxu q_x; Hux Huu q_u] entry (i, j), i = g+4r, as one LDS word per stage (offset + k*stride):
            // Hxx: packed HXX[k];  Hux: only (0,2) is non-zero (HUX02[k]);  Huu: constant R2+W2 off the diagonal, HUUD[k] on it;
            // column NV: gradient QXU[k].  Stage N-1 adds the dense rank-one blocks HUXL / HUUL.
            // (i, j below are already in the numbering of the stage arrays; the gradient sits in row AND column NV of the
            //  symmetric homogeneous form)
            const int i = lg(g + 4 * r), j = jl;
            unsigned off = (unsigned)L.CV, stride = 0, last = (unsigned)L.CV;
            if (i < NX && j < NX) {
                off = (unsigned)(L.HXX + (i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i)); stride = NXX;
                ls.h_m |= (1u << r) | (i >= j ? (16u << r) : 0u);
            } else if (i >= NX && i < NV && j < NX) {
                if (i == NX && j == 2) { off = (unsigned)L.HUX02; stride = 1; }
                last = (unsigned)(L.HUXL + (i - NX) * NX + j);
            } else if (i >= NX && i < NV && j >= NX && j < NV) {
                const int a = i - NX, b = j - NX;
                if (a == b) { off = (unsigned)(L.HUUD + a); stride = NU; }
                else off = (unsigned)(L.CST + MMPC_C_RW2 + a * NU + b);
                last = (unsigned)(L.HUUL + (a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a));
            } else if (i < NV && j == NV) {
                off = (unsigned)(L.QXU + i); stride = NV;
                if (i < NX) ls.h_m |= (1u << r) | (16u << r);
            } else if (i == NV && j < NX) {   // transposed gradient: part of the cost-to-go, not stored a second time
                off = (unsigned)(L.QXU + j); stride = NV;
                ls.h_m |= (1u << r);
            }
            ls.h_o[r] = off | (stride << 16);
            ls.h_l[r] = last;
            ls.p_o[r] = ((ls.h_m >> (4 + r)) & 1u) ? ls.h_o[r] : (unsigned)(L.DUMP + lane);
        }
#pragma unroll
        for (int l = 0; l < F::NLEG; l++) {
            // pivot row of input a = row NX+1+a of the stage matrix, held by lane group (NX+1+a) & 3: its entry in column j,
            // divided by the pivot, is a gain-row entry (j < NX), the feed-forward (j = NX) or the coupling to a later input.
            // A leg eliminates input a0 or the pair (a0, a0 + 1), whose rows sit in neighbouring lane groups
            const int a0 = F::PAIRS ? 2 * l : l;
            unsigned off = (unsigned)((GK ? O_KDUMP : L.DUMP) + lane), stride = 0;
#pragma unroll
            for (int a = a0; a < NU && a <= a0 + (F::PAIRS ? 1 : 0); a++) {
                const int ta = NX + 1 + a;
                if (g == (ta & 3)) {
                    if (j < NX) { off = (unsigned)((GK ? O_KK : L.KK) + a * NX + j); stride = NU * NX; }
                    else if (j == NX) { off = (unsigned)((GK ? O_KF : L.KF) + a); stride = NU; }
                    else if (j > ta && j <= NV) { const int b2 = j - NX - 1; off = (unsigned)((GK ? O_KU : L.KU) + a * (2 * NU - a - 1) / 2 + (b2 - a - 1)); stride = NPU; }
                }
            }
            ls.kl_b[l] = (int)off * 8; ls.kl_s[l] = (int)stride * 8;
        }
        {
            unsigned fv = 0, fx = 0;
            if (lane < NX) {
                // term 0 of the row tables is the diagonal (coefficient 1); the others sit in columns 2..5 or in an input column
                unsigned cu = 0;
                for (int q = 1; q < 5; q++) {
                    const int col = TB::rcol(lane, q), id = TB::rcv(lane, q);
                    if (id == 0) continue;
                    if (col >= NX) cu = (unsigned)id; else fv |= (unsigned)id << (8 * (col - 2));
                }
                const int a = (lane == 3 || lane == 4) ? 0 : (lane >= 5 ? lane - 4 : -1);
                fx = (cu << 8) | ((unsigned)(a + 1) << 16);
            }
            ls.f_v = fv; ls.f_x = fx;
        }
    }
    LANES_END

    const bool diagw = true;  // dense weights are routed to the generic path by the host
    (void)diagw;
    // bounds of pair idx (k,v): returns activity of each side
    auto pair_bounds = [&](int k, int v, double &lo, double &hi, bool &alo, bool &ahi) {
        if (v < NX) {
            lo = CST[MMPC_C_XLIM + v]; hi = CST[MMPC_C_XLIM + 9 + v];
            const bool ex = k >= 1;
            alo = ex && mmpc_finite(lo); ahi = ex && mmpc_finite(hi);
        } else {
            const int a = v - NX;
            const double ul = ulast_at(k, a);
            lo = mmpc_vmax(CST[MMPC_C_ULIM + a], ul + CST[MMPC_C_DULIM + a]);
            hi = mmpc_vmin(CST[MMPC_C_ULIM + 5 + a], ul + CST[MMPC_C_DULIM + 5 + a]);
            const bool ex = k < N;
            alo = ex && mmpc_finite(lo); ahi = ex && mmpc_finite(hi);
        }
    };
    // cost weight (diagonal, already doubled) of variable v at stage k;  second = W2 weight (inputs)
    auto w_diag = [&](int k, int v) -> double {
        if (v < NX) return CST[(k < N ? MMPC_C_WQ : MMPC_C_WP) + v];
        return k < N ? CST[MMPC_C_WR + v - NX] : 0.0;
    };

    double mu = P.mu_init;
    // ------------------------------------------------------------------ bound push of the initial point (own phase: the
    //                                                                    stage lanes below read what the pair lanes move)
    LANES_BEGIN
    for (int idx = lane; idx < NPAIR; idx += MMPC_WAVE) {
        const int k = idx / NV, v = idx % NV;
        double lo, hi; bool alo, ahi;
        pair_bounds(k, v, lo, hi, alo, ahi);
        if (alo || ahi) XU[idx] = mmpc_bound_push(XU[idx], alo ? lo : -INFINITY, ahi ? hi : INFINITY);
    }
    LANES_END
    // ------------------------------------------------------------------ slack / multiplier init
    LANES_BEGIN
    auto &ls = MMPC_LS;
#pragma unroll
    for (int p = 0; p < NPASS; p++) {
        mmpc_sched_fence();
        const int idx = lane + MMPC_WAVE * p;
        ls.lo_z[p] = 0.0; ls.hi_z[p] = 0.0; ls.b_lo[p] = -1e300; ls.b_hi[p] = 1e300;
        if (idx < NPAIR) {
            const int k = idx / NV, v = idx % NV;
            double lo, hi; bool alo, ahi;
            pair_bounds(k, v, lo, hi, alo, ahi);
            const double val = XU[idx];
            if (alo) { ls.b_lo[p] = lo; ls.lo_z[p] = mu / mmpc_box_t(val - lo); }
            if (ahi) { ls.b_hi[p] = hi; ls.hi_z[p] = mu / mmpc_box_t(hi - val); }
        }
    }
#pragma unroll
    for (int r = 0; r < MCR; r++) { ls.ct[r] = 1.0; ls.cz[r] = 0.0; ls.cdt[r] = 0.0; }
    {
        MMPC_ROW_LANE
        if (rlane) {
            const double px = XU[rk * NV], py = XU[rk * NV + 1], sk = S[rk];
#pragma unroll
            for (int r = 0; r < MCR; r++) {
                const int m = rs + RG * r;
                if (m < M) {
                    const double *o = obs_ptr(rk, m);
                    const double dx = px - o[0], dy = py - o[1];
                    const double h = (o[2] + MMPC_BASE_R) - sqrt(dx * dx + dy * dy) - sk;
                    ls.ct[r] = mmpc_vmax(-h, 1e-2); ls.cz[r] = mu / ls.ct[r];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { ls.st[i] = 1.0; ls.sz[i] = 0.0; }
    if (lane < NS) {
        const int k = lane;
        const double *xk = XU + k * NV;
        if (NSELF) {
            double dr[3], dz[3], sn, cs;
            mmpc_sincos(xk[2], &sn, &cs);
            mmpc_arm_segments_fast(xk[NX - 3], xk[NX - 2], xk[NX - 1], dr, dz);
#pragma unroll
            for (int i = 0; i < NSELF; i++) {
                const double h = mmpc_self_row(i, xk[0], xk[1], cs, sn, dr, dz, nullptr) - S[slack_idx(k)];
                ls.st[i] = mmpc_vmax(-h, 1e-2); ls.sz[i] = mu / ls.st[i];
            }
        }
    }
    // count active rows (for the KKT error scaling)
    double cnt = 0.0;
#pragma unroll
    for (int p = 0; p < NPASS; p++) cnt += (ls.lo_z[p] != 0.0 ? 1.0 : 0.0) + (ls.hi_z[p] != 0.0 ? 1.0 : 0.0);
    if (lane < NS) cnt += (double)(M + NSELF);
    MMPC_WR(0) = cnt;
    LANES_END
    const double nrows_act = MMPC_RED_SUM(0);
    const double inv_rows = 1.0 / (nrows_act + (double)(NS * NX)), inv_nrows = 1.0 / (nrows_act > 0.0 ? nrows_act : 1.0);

    int status = 1, it = 0, nfilt = 0, filt_init = 0;
    double E0 = 0.0, th_max = 0.0, th_min = 0.0;
    // save area of a suspended solve (layout: mmpc_fast_state_doubles)
    constexpr int MCS = MC > 0 ? MC : 1, NLREG = 2 * NPASS + 2 * MCS + 8;
    // (offsets into io.state; the pointers are formed where they are used so that nothing of this stays live in the loop)
    constexpr int ST_S = NPAIR, ST_LAM = NPAIR + NS, ST_FILT = NPAIR + NS + NS * NX, ST_SCAL = ST_FILT + 2 * MMPC_FCAP,
                  ST_LANE = ST_SCAL + MMPC_NSCAL;
    int nsmall_r = 0;
    double prox_r = 0.0, delta_r = 0.0;
    if (CONT && io.resume) {
        // ---- continue a suspended solve: the state the uninterrupted loop would hold at this point
        const double *const st_xu = io.state, *const st_s = io.state + ST_S, *const st_lam = io.state + ST_LAM,
                     *const st_filt = io.state + ST_FILT, *const st_scal = io.state + ST_SCAL, *const st_lane = io.state + ST_LANE;
        LANES_BEGIN
        auto &ls = MMPC_LS;
        for (int i = lane; i < NPAIR; i += MMPC_WAVE) XU[i] = st_xu[i];
        for (int i = lane; i < NS; i += MMPC_WAVE) S[i] = st_s[i];
        for (int i = lane; i < NS * NX; i += MMPC_WAVE) LAM[i] = st_lam[i];
        for (int i = lane; i < 2 * MMPC_FCAP; i += MMPC_WAVE) FILT[i] = st_filt[i];
        const double *q = st_lane + lane * NLREG;
#pragma unroll
        for (int p = 0; p < NPASS; p++) { ls.lo_z[p] = q[p]; ls.hi_z[p] = q[NPASS + p]; }
#pragma unroll
        for (int m = 0; m < MCR; m++) { ls.ct[m] = q[2 * NPASS + m]; ls.cz[m] = q[2 * NPASS + MCS + m]; }
#pragma unroll
        for (int i = 0; i < 4; i++) { ls.st[i] = q[2 * NPASS + 2 * MCS + i]; ls.sz[i] = q[2 * NPASS + 2 * MCS + 4 + i]; }
        LANES_END
        mu = st_scal[0]; th_max = st_scal[1]; th_min = st_scal[2]; prox_r = st_scal[3];
        it = (int)st_scal[4]; nfilt = (int)st_scal[5]; filt_init = (int)st_scal[6]; nsmall_r = (int)st_scal[7]; delta_r = st_scal[8];
    }
    const int it_start = CONT ? it : 0;
    // results of the evaluation of the current point (iterate or line-search trial)
    double err_d = 0.0, err_p = 0.0, tzmax = 0.0, tzmin = 0.0, zsum = 0.0, cost_c = 0.0, th_c = 0.0, sumlog = 0.0;
    // line-search state: the evaluation of a trial point IS the evaluation the next iteration starts from (98.6 % of the
    // first trials are accepted), so the loop below evaluates once per trial and never a second time for the accepted one
    int in_ls = 0, lspass = 0, lsi = 0, nsmall = nsmall_r;
    double prox = prox_r;   // proximal term for crawling iterations (mmpc_prox_update)
    double delta_last = delta_r;   // last successful inertia correction
    double alpha = 0.0, ap = 1.0, ad = 1.0, dphi = 0.0, phi0 = 0.0, th0 = 0.0;

    // ---- move to a trial point: the primal variables, the equality multipliers and the slacks of the nonlinear rows by
    //      d_alpha times the direction (signed: a rejected trial is moved back by the difference); with `first` also the
    //      row multipliers by alpha_d (they do not depend on alpha; evaluated with the slacks of the point the step starts at)
    auto apply_step = [&](double d_alpha, bool first) {
        LANES_BEGIN
        auto &ls = MMPC_LS;
        {
            MMPC_ROW_LANE
            if (rlane) {
#pragma unroll
                for (int r = 0; r < MCR; r++) {
                    if (RG == 1 || rs + RG * r < M) {
                        const double t = ls.ct[r], z = ls.cz[r], dtv = ls.cdt[r], tn = t + d_alpha * dtv;
                        if (first) { const double it_ = mmpc_rcp(t); ls.cz[r] = mmpc_z_safeguard_fast(z + ad * (mu * it_ - z - z * it_ * dtv), tn, mu); }
                        ls.ct[r] = tn;
                    }
                }
            }
        }
        if (lane < NS) {
#pragma unroll
            for (int i = 0; i < NSELF; i++) {
                const double t = ls.st[i], z = ls.sz[i], dtv = ls.sdt[i], tn = t + d_alpha * dtv;
                if (first) { const double it_ = mmpc_rcp(t); ls.sz[i] = mmpc_z_safeguard_fast(z + ad * (mu * it_ - z - z * it_ * dtv), tn, mu); }
                ls.st[i] = tn;
            }
        }
        // (same phase: every lane touches only its own registers and its own words of XU / LAM / S)
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            mmpc_sched_fence();
            const int idx = lane + MMPC_WAVE * p;
            if (idx < NPAIR) {
                const double val = XU[idx], dv = DXU[idx];
                if (first) {
                    const double lo = ls.b_lo[p], hi = ls.b_hi[p];
                    const bool alo = lo > -1e299, ahi = hi < 1e299;
                    const double vn = val + d_alpha * dv;
                    if (alo) { const double z = ls.lo_z[p], it_ = mmpc_rcp(mmpc_box_t(val - lo)); ls.lo_z[p] = mmpc_z_safeguard_fast(z + ad * (mu * it_ - z - z * it_ * dv), mmpc_box_t(vn - lo), mu); }
                    if (ahi) { const double z = ls.hi_z[p], it_ = mmpc_rcp(mmpc_box_t(hi - val)); ls.hi_z[p] = mmpc_z_safeguard_fast(z + ad * (mu * it_ - z + z * it_ * dv), mmpc_box_t(hi - vn), mu); }
                }
                if (idx >= NX) XU[idx] = val + d_alpha * dv;   // x_0 is data
            }
        }
        for (int i = lane; i < NS * NX; i += MMPC_WAVE) if (i >= NX) LAM[i] += d_alpha * DLAM[i];
        for (int i = lane; i < NS; i += MMPC_WAVE) S[i] += d_alpha * DS[i];
        LANES_END
    };

    // ---- second-order correction (Waechter & Biegler 2006, section 2.4; algorithm and deviations: mmpc_core.h).  This kernel moves to
    //      a trial point in place, so a correction is a small state machine around the evaluation:
    //        soc_st 0  no correction in progress
    //               1  back at x_k, evaluated again (with the multipliers' step taken): the constraint residuals are replaced by
    //                  c_soc = a c_soc + c(trial) and the direction is formed from them (corrected pass)
    //               2  the corrected trial x_k + a_soc d_soc has been evaluated: filter test with the length / slope of the step it corrects
    //               3  the correction failed: back at x_k, evaluated again, the uncorrected direction restored from io.soc - only the
    //                  row steps (registers) are formed again, then the line search goes on with alpha / 2
    //      What has to survive a corrected pass lives in this instance's block of global memory (rare path: no LDS, no registers):
    constexpr int NRS = (MC > 0 ? MC : 0) + NSELF;                                     // rows per stage that are not box rows
    constexpr int O_D = 0, O_TC = O_D + NS * NV + NS + NS * NX, O_TR = O_TC + NS * NX,  // direction | residuals of the trial point
                  O_AC = O_TR + NS * NRS, O_AR = O_AC + NS * NX;                         // running c_soc (dynamics | rows)
    static_assert(O_AR + NS * NRS <= mmpc_soc_doubles(N, NX, NU, NRS), "second-order correction scratch");
    int soc_st = 0, soc_p = 0, fatal = 0;
    double a_soc = 0.0, alpha0 = 0.0, a_prev = 0.0, th_prev = 0.0;
#ifdef MMPC_EMU
#define MMPC_SOC_FENCE()
#else
#define MMPC_SOC_FENCE() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#endif
    // the uncorrected direction to / from the scratch
    auto soc_dir = [&](bool save) {
        LANES_BEGIN
        double *q = io.soc + O_D;
        for (int i = lane; i < NS * NV; i += MMPC_WAVE) { if (save) q[i] = DXU[i]; else DXU[i] = q[i]; }
        for (int i = lane; i < NS; i += MMPC_WAVE) { if (save) q[NS * NV + i] = DS[i]; else DS[i] = q[NS * NV + i]; }
        for (int i = lane; i < NS * NX; i += MMPC_WAVE) { if (save) q[NS * NV + NS + i] = DLAM[i]; else DLAM[i] = q[NS * NV + NS + i]; }
        LANES_END
        MMPC_SOC_FENCE()
    };
    // constraint residuals of the point that has just been evaluated (dynamics: CD; rows: h + t, re-derived from the evaluation's
    // caches).  acc = false: stored as the residuals of a rejected trial.  acc = true (back at x_k): c_soc <- a_prev c_soc + c(trial),
    // c_soc = c(x_k) in the first round; the dynamics part replaces CD, the row part is read by the corrected pass, and the
    // terminal self rows' sum of w (h + t) - formed by the evaluation, quirk Q1 - is redone with it.
    auto soc_rows = [&](bool acc, bool first) {
        LANES_BEGIN
        auto &ls = MMPC_LS;
        double *tc = io.soc + O_TC, *tr = io.soc + O_TR, *ac = io.soc + O_AC, *ar = io.soc + O_AR;
        for (int i = lane; i < N * NX; i += MMPC_WAVE) {
            if (!acc) tc[i] = CD[i];
            else { const double v = a_prev * (first ? CD[i] : ac[i]) + tc[i]; ac[i] = v; CD[i] = v; }
        }
        {
            MMPC_ROW_LANE
            if (rlane && (RG > 1 || lane < NS)) {
                const double px = XU[rk * NV], py = XU[rk * NV + 1], sk = S[rk];
#pragma unroll
                for (int r = 0; r < MCR; r++) {
                    const int m = rs + RG * r;
                    if (m < M) {
                        const double *o = obs_ptr(rk, m);
                        const double dx = px - o[0], dy = py - o[1];
                        const double rr = (o[2] + MMPC_BASE_R) - sqrt(dx * dx + dy * dy) - sk + ls.ct[r];
                        const int e = rk * NRS + m;
                        if (!acc) tr[e] = rr; else ar[e] = a_prev * (first ? rr : ar[e]) + tr[e];
                    }
                }
            }
        }
        if (NSELF && lane < NS) {
            const int k = lane;
            const double sn = TRG[k * 8], cs = TRG[k * 8 + 1], sks = S[slack_idx(k)];
            double dr[3], dz[3], swr = 0.0;
#pragma unroll
            for (int a = 0; a < 3; a++) { dr[a] = TRG[k * 8 + 2 + a]; dz[a] = TRG[k * 8 + 5 + a]; }
#pragma unroll
            for (int i = 0; i < NSELF; i++) {
                const double rr = mmpc_self_row(i, XU[k * NV], XU[k * NV + 1], cs, sn, dr, dz, nullptr) - sks + ls.st[i];
                const int e = k * NRS + M + i;
                if (!acc) tr[e] = rr;
                else { const double v = a_prev * (first ? rr : ar[e]) + tr[e]; ar[e] = v; swr += ls.sz[i] * mmpc_rcp(ls.st[i]) * v; }
            }
            if (acc && k == N) SN[2] = swr;
        }
        LANES_END
        MMPC_SOC_FENCE()
    };

    MMPC_T0()
#pragma unroll 1
    for (;;) {
        MMPC_TS(0)
        // ============================================================ E1 (circle rows where RG lanes share a stage)
        if (RG > 1 && M > 0) {
            LANES_BEGIN
            auto &ls = MMPC_LS;
            double e_p = 0.0, tzmax = 0.0, tzmin = 1e300, zsum = 0.0, th = 0.0, rb0 = 0.0, rb1 = 0.0, rz = 0.0;
            MmpcLogAcc la; la.init();
            MMPC_ROW_LANE
            if (rlane) {
                const double px = XU[rk * NV], py = XU[rk * NV + 1], sk = S[rk];
                double ob[MCR * 3];
#pragma unroll
                for (int r = 0; r < MCR; r++) {
                    const double *o = obs_ptr(rk, rs + RG * r < M ? rs + RG * r : 0);
                    ob[3 * r] = o[0]; ob[3 * r + 1] = o[1]; ob[3 * r + 2] = o[2];
                }
#pragma unroll
                for (int r = 0; r < MCR; r++) {
                    if (rs + RG * r < M) {
                        const double dx = px - ob[3 * r], dy = py - ob[3 * r + 1], m2 = dx * dx + dy * dy, id = mmpc_rsqrt(m2), d = m2 * id;
                        const double nxv = dx * id, nyv = dy * id;
                        const double h = (ob[3 * r + 2] + MMPC_BASE_R) - d - sk;
                        const double t = ls.ct[r], z = ls.cz[r];
                        rb0 -= nxv * z; rb1 -= nyv * z; rz += z;
                        e_p = mmpc_vmax(e_p, fabs(h + t)); th += fabs(h + t); la.mul(t);
                        tzmax = mmpc_vmax(tzmax, t * z); tzmin = mmpc_vmin(tzmin, t * z); zsum += z;
                    }
                }
                // the stage lane (rs = 0) keeps its sums in registers, the others hand theirs over through the stage's Hessian block
                if (rs > 0) { double *cpo = HXX + rk * NXX + (rs - 1) * 3; cpo[0] = rb0; cpo[1] = rb1; cpo[2] = rz; }
            }
            ls.cp[0] = rb0; ls.cp[1] = rb1; ls.cp[2] = rz;
            MMPC_WR(1) = e_p; MMPC_WR(2) = tzmax; MMPC_WR(3) = tzmin; MMPC_WR(4) = zsum; MMPC_WR(6) = th; MMPC_WR(7) = la.mant; MMPC_WR(0) = (double)la.ex;
            LANES_END
        }
        // ============================================================ E1 (stage lanes)
        LANES_BEGIN
        auto &ls = MMPC_LS;
        double e_p = 0.0, tzmax = 0.0, tzmin = 1e300, zsum = 0.0, zeq = 0.0, phi = 0.0, th = 0.0, slog = 0.0;
        MmpcLogAcc la; la.init();
        if (RG > 1 && M > 0) {
            e_p = MMPC_WR(1); tzmax = MMPC_WR(2); tzmin = MMPC_WR(3); zsum = MMPC_WR(4); th = MMPC_WR(6); la.mant = MMPC_WR(7); la.ex = (int)MMPC_WR(0);
        }
        if (lane < NS) {
            const int k = lane, k1 = k < N ? k + 1 : k;
            // every LDS word of this stage first (the stores below would otherwise pin each later load behind them)
            // (ob: the obstacles of this stage where the stage lane owns all circle rows, else the sums of the other lanes' rows)
            double xk[NV], xn1[NX], ln[NX], rb[NV], ob[RG > 1 ? (RG - 1) * 3 : (MC > 0 ? MC : 1) * 3];
#pragma unroll
            for (int j = 0; j < NV; j++) { xk[j] = XU[k * NV + j]; rb[j] = 0.0; }
#pragma unroll
            for (int j = 0; j < NX; j++) { xn1[j] = XU[k1 * NV + j]; ln[j] = LAM[k1 * NX + j]; rb[j] = k >= 1 ? LAM[k * NX + j] : 0.0; }
            if (RG > 1) {
#pragma unroll
                for (int q = 0; q < (RG - 1) * 3; q++) ob[q] = M > 0 ? HXX[k * NXX + q] : 0.0;
            } else {
#pragma unroll
                for (int m = 0; m < M; m++) { const double *o = obs_ptr(k, m); ob[3 * m] = o[0]; ob[3 * m + 1] = o[1]; ob[3 * m + 2] = o[2]; }
            }
            const double sk = S[k], sks_ld = S[slack_idx(k)];
            mmpc_sched_fence();
            double sn, cs;
            mmpc_sincos(xk[2], &sn, &cs);
            double *cv = CV + k * MMPC_NCV;
            if (k < N) {
                const double *uk = xk + NX;
                const double a32 = -dt * uk[0] * sn, a42 = dt * uk[0] * cs, a43 = dt * xk[5], a34 = -dt * xk[5],
                             a35 = -dt * xk[4], a45 = dt * xk[3], b30 = dt * cs, b40 = dt * sn;
                cv[0] = 0.0; cv[1] = 1.0; cv[2] = dt; cv[3] = a32; cv[4] = a42; cv[5] = a43; cv[6] = a34; cv[7] = a35;
                cv[8] = a45; cv[9] = b30; cv[10] = b40;
                double c[NX];
                c[0] = xk[0] + dt * xk[3] - xn1[0]; c[1] = xk[1] + dt * xk[4] - xn1[1]; c[2] = xk[2] + dt * xk[5] - xn1[2];
                c[3] = xk[3] + dt * (uk[0] * cs - xk[4] * xk[5]) - xn1[3];
                c[4] = xk[4] + dt * (uk[0] * sn + xk[3] * xk[5]) - xn1[4];
                c[5] = xk[5] + dt * uk[1] - xn1[5];
                if (KIND == 0) { c[6] = xk[6] + dt * uk[2] - xn1[6]; c[7] = xk[7] + dt * uk[3] - xn1[7]; c[8] = xk[8] + dt * uk[4] - xn1[8]; }
#pragma unroll
                for (int j = 0; j < NX; j++) { CD[k * NX + j] = c[j]; e_p = mmpc_vmax(e_p, fabs(c[j])); th += fabs(c[j]); zeq += fabs(ln[j]); }
                // - A^T lam_{k+1}, - B^T lam_{k+1}  (sparse, base.py:19-26)
                rb[0] -= ln[0]; rb[1] -= ln[1];
                rb[2] -= ln[2] + a32 * ln[3] + a42 * ln[4];
                rb[3] -= ln[3] + dt * ln[0] + a43 * ln[4];
                rb[4] -= ln[4] + dt * ln[1] + a34 * ln[3];
                rb[5] -= ln[5] + dt * ln[2] + a35 * ln[3] + a45 * ln[4];
                rb[NX + 0] -= b30 * ln[3] + b40 * ln[4];
                rb[NX + 1] -= dt * ln[5];
                if (KIND == 0) {
                    rb[6] -= ln[6]; rb[7] -= ln[7]; rb[8] -= ln[8];
                    rb[NX + 2] -= dt * ln[6]; rb[NX + 3] -= dt * ln[7]; rb[NX + 4] -= dt * ln[8];
                }
            }
            double rds = 2 * Sw * sk, selfz = 0.0;
            phi += Sw * sk * sk;
            if (RG > 1 && M > 0) {   // circle rows of this stage: own rows first, then the other lanes' sums in lane order
                double c0 = ls.cp[0], c1 = ls.cp[1], c2 = ls.cp[2];
#pragma unroll
                for (int g = 0; g < RG - 1; g++) { c0 += ob[3 * g]; c1 += ob[3 * g + 1]; c2 += ob[3 * g + 2]; }
                rb[0] += c0; rb[1] += c1; rds -= c2;
            }
#pragma unroll
            for (int m = 0; m < (RG > 1 ? 0 : M); m++) {
                const double dx = xk[0] - ob[3 * m], dy = xk[1] - ob[3 * m + 1], m2 = dx * dx + dy * dy, id = mmpc_rsqrt(m2), d = m2 * id;
                const double nxv = dx * id, nyv = dy * id;
                const double h = (ob[3 * m + 2] + MMPC_BASE_R) - d - sk;
                const double t = ls.ct[m], z = ls.cz[m];
                rb[0] -= nxv * z; rb[1] -= nyv * z; rds -= z;
                e_p = mmpc_vmax(e_p, fabs(h + t)); th += fabs(h + t); la.mul(t);
                tzmax = mmpc_vmax(tzmax, t * z); tzmin = mmpc_vmin(tzmin, t * z); zsum += z;
            }
            if (NSELF) {
                double dr[3], dz[3];
                mmpc_arm_segments_fast(xk[NX - 3], xk[NX - 2], xk[NX - 1], dr, dz);
                TRG[k * 8 + 0] = sn; TRG[k * 8 + 1] = cs;
#pragma unroll
                for (int a = 0; a < 3; a++) { TRG[k * 8 + 2 + a] = dr[a]; TRG[k * 8 + 5 + a] = dz[a]; }
                const double sks = sks_ld;
                double sw = 0.0, sit = 0.0, swr = 0.0, swg[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < NSELF; i++) {
                    double g6[6];
                    const double h = mmpc_self_row(i, xk[0], xk[1], cs, sn, dr, dz, g6) - sks;
                    const double t = ls.st[i], z = ls.sz[i];
#pragma unroll
                    for (int a = 0; a < 6; a++) rb[mmpc_y(a)] += g6[a] * z;
                    selfz += z;
                    e_p = mmpc_vmax(e_p, fabs(h + t)); th += fabs(h + t); la.mul(t);
                    tzmax = mmpc_vmax(tzmax, t * z); tzmin = mmpc_vmin(tzmin, t * z); zsum += z;
                    if (k == N) {
                        const double it_ = mmpc_rcp(t), w = z * it_;
                        sw += w; sit += it_; swr += w * (h + t);
#pragma unroll
                        for (int a = 0; a < 6; a++) swg[a] += w * g6[a];
                    }
                }
                if (k == N) {   // terminal self rows belong to s_{N-1} (Q1): hand their sums to lane N-1
                    SN[0] = sw; SN[1] = sit; SN[2] = swr; SN[3] = selfz;
#pragma unroll
                    for (int a = 0; a < 6; a++) SN[4 + a] = swg[a];
                }
            }
            if (k < N) rds -= selfz;
            RDS[k] = rds;
#pragma unroll
            for (int j = 0; j < NV; j++) RB[k * NV + j] = rb[j];
        }
        if (RG > 1 || lane < NS) slog = la.value();
        MMPC_WR(1) = e_p; MMPC_WR(2) = tzmax; MMPC_WR(3) = tzmin; MMPC_WR(4) = zsum; MMPC_WR(5) = phi; MMPC_WR(6) = th; MMPC_WR(7) = slog;
        MMPC_WR(8) = zeq;   // |multipliers of the dynamics|: s_d counts them, s_c does not
        if (NSELF == 0 && lane == 0) { for (int a = 0; a < 10; a++) SN[a] = 0.0; }
        LANES_END
        MMPC_TS(1)
        // ============================================================ E1 (pair lanes) + s residual
        LANES_BEGIN
        auto &ls = MMPC_LS;
        double e_d = 0.0, e_p = MMPC_WR(1), tzmax = MMPC_WR(2), tzmin = MMPC_WR(3), zsum = MMPC_WR(4), phi = MMPC_WR(5), th = MMPC_WR(6);
        MmpcLogAcc la; la.init();
        // long horizons read the references and the previous inputs from HBM / L2: the loads of ALL passes ahead of the loop (one
        // exposed round trip instead of one per pass - the scheduler fences below keep the compiler from doing that itself)
        double g_ref[SLIM ? NPASS : 1], g_ul[SLIM ? NPASS : 1];
        if (SLIM) {
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int idx = lane + MMPC_WAVE * p, ii = idx < NPAIR ? idx : 0;
                const int k = ii / NV, v = ii % NV;
                const bool isu = v >= NX && k < N;
                g_ref[p] = ref_at(ii, k, v);
                g_ul[p] = ulast_at(isu ? k : 0, isu ? v - NX : 0);
            }
        }
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            mmpc_sched_fence();
            const int idx = lane + MMPC_WAVE * p;
            if (idx < NPAIR) {
                const int k = idx / NV, v = idx % NV;
                const double lo = ls.b_lo[p], hi = ls.b_hi[p];
                const bool alo = lo > -1e299, ahi = hi < 1e299;
                // every LDS word this pair needs, ahead of the arithmetic and of the per-side blocks (one round trip per pass);
                // the input-only terms read a valid dummy address for state variables and carry weight 0 there
                const bool isu = v >= NX && k < N;
                const int au = isu ? v - NX : 0;
                const double val = XU[idx], ref = SLIM ? g_ref[SLIM ? p : 0] : ref_at(idx, k, v), rb0 = RB[idx];
                const double ul = SLIM ? g_ul[SLIM ? p : 0] : ulast_at(isu ? k : 0, au), ww0 = CST[MMPC_C_WW + au];
                const double wq = CST[v < NX ? (k < N ? MMPC_C_WQ : MMPC_C_WP) + v : MMPC_C_WR + v - NX];
                mmpc_sched_fence();
                // cost gradient / value (diagonal weights): mpc_wholebody_qref.py:192-201,240-242
                double e = val - ref;
                if (KIND == 1 && v == 2) e = mmpc_angle_diff(val, ref);
                const double wqe = (v < NX || k < N) ? wq : 0.0, ww = isu ? ww0 : 0.0, e2 = val - ul;
                double g = wqe * e + ww * e2;
                phi += 0.5 * wqe * e * e + 0.5 * ww * e2 * e2;
                double r = rb0 + g;
                {   // both sides without branches: an absent side has z = 0 and is given t = 1 (log 1 = 0)
                    const double tl = alo ? mmpc_box_t(val - lo) : 1.0, zl = ls.lo_z[p], th_ = ahi ? mmpc_box_t(hi - val) : 1.0, zh = ls.hi_z[p];
                    const double pl = tl * zl, ph = th_ * zh;
                    r += zh - zl; la.mul(tl); la.mul(th_);
                    tzmax = mmpc_vmax(tzmax, mmpc_vmax(pl, ph));
                    tzmin = mmpc_vmin(tzmin, mmpc_vmin(alo ? pl : 1e300, ahi ? ph : 1e300));
                    zsum += zl + zh;
                }
                RB[idx] = g;   // keep the plain cost gradient for the assembly / directional derivative
                const bool isvar = v < NX ? (k >= 1) : (k < N);
                if (isvar) e_d = mmpc_vmax(e_d, fabs(r));
            }
        }
        MMPC_WR(7) += la.value();   // sum of log t over all rows (the barrier term is applied after the mu update)
        if (lane < NS) e_d = mmpc_vmax(e_d, fabs(RDS[lane] - (lane == N - 1 ? SN[3] : 0.0)));
        MMPC_WR(0) = e_d; MMPC_WR(1) = e_p; MMPC_WR(2) = tzmax; MMPC_WR(3) = tzmin; MMPC_WR(4) = zsum; MMPC_WR(5) = phi; MMPC_WR(6) = th;
        LANES_END
        err_d = MMPC_RED_MAX(0); err_p = MMPC_RED_MAX(1); tzmax = MMPC_RED_MAX(2); tzmin = MMPC_RED_MIN(3);
        zsum = MMPC_RED_SUM(4); cost_c = MMPC_RED_SUM(5); th_c = MMPC_RED_SUM(6); sumlog = MMPC_RED_SUM(7);
        const double zeq_c = MMPC_RED_SUM(8);
        if (in_ls) {
            // ---- filter test of the trial point just evaluated (Waechter-Biegler; + filter reset heuristic)
            const double phi = cost_c - mu * sumlog, th = th_c;
            bool okf = th < th_max;
            for (int i = 0; i < nfilt && okf; i++) if (th >= FILT[2 * i] && phi >= FILT[2 * i + 1]) okf = false;
            const double a_t = soc_st == 2 ? alpha0 : alpha;   // (a corrected step is tested with the length of the step it corrects)
            const bool ftype = dphi < 0 && th0 <= th_min && a_t * mmpc_powf(-dphi, 2.3f) > mmpc_powf(th0, 1.1f);
            bool accepted = false, augment = false;
            if (okf) {
                if (ftype) accepted = phi <= phi0 + 1e-8 * a_t * dphi + 1e-14 * fabs(phi0);
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
            if (soc_st == 2) {
                if (accepted) { soc_st = 0; alpha = a_soc; }
                else if (th > 0.99 * th_prev || soc_p + 1 >= MMPC_SOC_MAX) {
                    // the correction failed: back to x_k, the uncorrected direction restored, evaluated there again (state 3)
                    apply_step(-a_soc, false);
                    soc_dir(false);
                    soc_st = 3; in_ls = 0;
                    continue;
                } else {
                    soc_rows(false, false);
                    apply_step(-a_soc, false);
                    soc_p++; a_prev = a_soc; th_prev = th; soc_st = 1; in_ls = 0;
                    continue;
                }
            } else if (!accepted && MMPC_SOC_MAX > 0 && lsi == 0 && lspass == 0 && io.soc && th >= th0 && th0 <= th_min) {
                // first trial rejected without reducing the infeasibility, theta(x_k) <= theta_min: second-order correction
                soc_dir(true);
                soc_rows(false, false);
                apply_step(-alpha, false);
                alpha0 = alpha; a_prev = alpha; th_prev = th; soc_p = 0; soc_st = 1; in_ls = 0;
                continue;
            }
            if (!accepted) {
                double anext = alpha;
                bool retry = false;
                if (lsi < MMPC_MAX_LS - 1) { anext = 0.5 * alpha; lsi++; retry = true; }
                else if (lspass == 0 && nfilt > 0) { nfilt = 0; lspass = 1; lsi = 0; anext = ap; retry = true; }   // filter reset heuristic
                if (retry) { apply_step(anext - alpha, false); alpha = anext; continue; }
            }
            in_ls = 0;   // accepted (or every trial rejected: the last one is kept, as IPOPT without restoration would stall too)
            mmpc_prox_update(ap, alpha, prox, nsmall);
            MMPC_TS(12)
        }
        if (soc_st == 1) soc_rows(true, soc_p == 0);
        if (soc_st == 0) {
        // IPOPT's termination test (Waechter & Biegler 2006, eq. 5-6): stationarity over s_d (all multipliers), complementarity over
        // s_c (the row multipliers alone)
        double sd = (zsum + zeq_c) * inv_rows, sc = zsum * inv_nrows;
        sd = (sd > 100.0 ? sd : 100.0) * 0.01;
        sc = (sc > 100.0 ? sc : 100.0) * 0.01;
        const double isd = mmpc_rcp(sd), isc = mmpc_rcp(sc);
        E0 = mmpc_vmax(mmpc_vmax(err_d * isd, err_p), tzmax * isc);
        if (!mmpc_finite(E0) || !mmpc_finite(th_c + cost_c + zsum + zeq_c)) {   // NaN or inf anywhere in the point or its data (the sums carry them)
#ifdef MMPC_EMU_DEBUG
            fprintf(stderr, "E0 nan it %d: err_d %g err_p %g tzmax %g tzmin %g zsum %g sumlog %g\n", it, err_d, err_p, tzmax, tzmin, zsum, sumlog);
#endif
            status = 2; break; }
        if (E0 <= tol) { status = 0; break; }
        if (it == P.max_iter) break;
        if (CONT && io.budget > 0 && it - it_start >= io.budget && io.state) {
            // ---- iteration budget of this launch used up: park the solve (the point has just been evaluated and is not
            //      converged; a resumed launch re-evaluates it and continues with the barrier update below)
            double *const st_xu = io.state, *const st_s = io.state + ST_S, *const st_lam = io.state + ST_LAM,
                   *const st_filt = io.state + ST_FILT, *const st_scal = io.state + ST_SCAL, *const st_lane = io.state + ST_LANE;
            LANES_BEGIN
            auto &ls = MMPC_LS;
            for (int i = lane; i < NPAIR; i += MMPC_WAVE) st_xu[i] = XU[i];
            for (int i = lane; i < NS; i += MMPC_WAVE) st_s[i] = S[i];
            for (int i = lane; i < NS * NX; i += MMPC_WAVE) st_lam[i] = LAM[i];
            for (int i = lane; i < 2 * MMPC_FCAP; i += MMPC_WAVE) st_filt[i] = i < 2 * nfilt ? FILT[i] : 0.0;
            double *q = st_lane + lane * NLREG;
#pragma unroll
            for (int p = 0; p < NPASS; p++) { q[p] = ls.lo_z[p]; q[NPASS + p] = ls.hi_z[p]; }
#pragma unroll
            for (int m = 0; m < MCR; m++) { q[2 * NPASS + m] = ls.ct[m]; q[2 * NPASS + MCS + m] = ls.cz[m]; }
#pragma unroll
            for (int i = 0; i < 4; i++) { q[2 * NPASS + 2 * MCS + i] = ls.st[i]; q[2 * NPASS + 2 * MCS + 4 + i] = ls.sz[i]; }
            if (lane == 0) {
                st_scal[0] = mu; st_scal[1] = th_max; st_scal[2] = th_min; st_scal[3] = prox;
                st_scal[4] = (double)it; st_scal[5] = (double)nfilt; st_scal[6] = (double)filt_init; st_scal[7] = (double)nsmall; st_scal[8] = delta_last;
            }
            LANES_END
            status = 3;   // MMPC_STATUS_SUSPENDED
            break;
        }
        {
            bool changed = false;
            for (;;) {
                const double compmu = mmpc_vmax(fabs(tzmax - mu), fabs(tzmin - mu));
                const double Emu = mmpc_vmax(mmpc_vmax(err_d * isd, err_p), compmu * isc);
                if (!(Emu <= 10.0 * mu && mu > tol / 10)) break;
                mu = mmpc_vmax(tol / 10, mmpc_vmin(0.2 * mu, mu * sqrt(mu)));
                changed = true;
            }
            if (changed) filt_init = 0;
        }
        phi0 = cost_c - mu * sumlog;   // barrier objective at the current point for the (possibly new) mu
        th0 = th_c;
        }   // soc_st == 0

        MMPC_TS(2)
        // ============================================================ Newton direction
        int failed = 0;
        double dw = 0.0;
        if (soc_st != 3) {   // (after a failed correction only the row steps D2 are formed again: the direction is back from io.soc)
#pragma unroll 1
        // (exact Lagrangian Hessian, + delta_w I by IPOPT's inertia correction where a pivot of the recursion is not positive - see mmpc_core.h)
        for (;;) {
            constexpr bool exact = true, dyn_curv = true;
            const double reg = prox + dw;
            int ric_bad = 0;   // a pivot of this pass was not positive (every lane factorises the same matrix: uniform)
            // ---- A1 (circle rows where RG lanes share a stage): w g g^T (+ exact curvature), gradient and s_k coupling of the
            //      lane's rows: cp = (Hxx, Hxy, Hyy, qx, qy, h_ss, g_ss, vx, vy)
            auto a1_rows = [&](auto TAG) {
            constexpr bool socm = decltype(TAG)::value; (void)socm;
            if (RG > 1 && M > 0) {
                LANES_BEGIN
                auto &ls = MMPC_LS;
                MMPC_ROW_LANE
                double cp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (rlane) {
                    const double px = XU[rk * NV], py = XU[rk * NV + 1], sk = S[rk];
                    double ob[MCR * 3];
#pragma unroll
                    for (int r = 0; r < MCR; r++) {
                        const double *o = obs_ptr(rk, rs + RG * r < M ? rs + RG * r : 0);
                        ob[3 * r] = o[0]; ob[3 * r + 1] = o[1]; ob[3 * r + 2] = o[2];
                    }
#pragma unroll
                    for (int r = 0; r < MCR; r++) {
                        if (rs + RG * r < M) {
                            const double ddx = px - ob[3 * r], ddy = py - ob[3 * r + 1], m2 = ddx * ddx + ddy * ddy, id = mmpc_rsqrt(m2), d = m2 * id;
                            const double g0 = -ddx * id, g1 = -ddy * id, hv = (ob[3 * r + 2] + MMPC_BASE_R) - d - sk;
                            const double t = ls.ct[r], z = ls.cz[r], it_ = mmpc_rcp(t), w = z * it_;
                            double res = hv + t;
                            if constexpr (socm) res = io.soc[O_AR + rk * NRS + rs + RG * r];   // corrected pass: (h + t)_soc
                            const double zh = mu * it_ + w * res;
                            cp[0] += w * g0 * g0; cp[1] += w * g1 * g0; cp[2] += w * g1 * g1;
                            if (exact) { const double zi = z * id; cp[0] -= zi * (1 - g0 * g0); cp[1] += zi * g0 * g1; cp[2] -= zi * (1 - g1 * g1); }
                            cp[3] += g0 * zh; cp[4] += g1 * zh;
                            cp[5] += w; cp[6] -= zh; cp[7] += w * g0; cp[8] += w * g1;
                        }
                    }
                    if (rs > 0) {
                        double *cpo = HXX + rk * NXX + (rs - 1) * 9;
#pragma unroll
                        for (int q = 0; q < 9; q++) cpo[q] = cp[q];
                    }
                }
#pragma unroll
                for (int q = 0; q < 9; q++) ls.cp[q] = cp[q];
                LANES_END
            }
            };
            if (soc_st == 1) a1_rows(MmpcTag<true>{}); else a1_rows(MmpcTag<false>{});
            // ---- A1 (stage lanes): stage Hessian incl. elimination of s_k
            auto a1_stage = [&](auto TAG) {
            constexpr bool socm = decltype(TAG)::value; (void)socm;
            LANES_BEGIN
            auto &ls = MMPC_LS;
            if (lane < NS) {
                const int k = lane, k1 = k < N ? k + 1 : k;
                // every LDS word of this stage first (one round trip instead of one per block below)
                double hxx[NXX], qx[NX], wd[NX], ob[RG > 1 ? (RG - 1) * 9 : (MC > 0 ? MC : 1) * 3];   // (ob: as in the evaluation)
#pragma unroll
                for (int j = 0; j < NX; j++) { wd[j] = CST[(k < N ? MMPC_C_WQ : MMPC_C_WP) + j]; qx[j] = RB[k * NV + j]; }
                const double *cv = CV + k * MMPC_NCV;
                const double l3 = LAM[k1 * NX + 3], l4 = LAM[k1 * NX + 4], cv3 = cv[3], cv4 = cv[4], cv9 = cv[9], cv10 = cv[10];
                const double px = XU[k * NV], py = XU[k * NV + 1], sk = S[k], sks = S[slack_idx(k)];
                if (RG > 1) {
#pragma unroll
                    for (int q = 0; q < (RG - 1) * 9; q++) ob[q] = M > 0 ? HXX[k * NXX + q] : 0.0;
                } else {
#pragma unroll
                    for (int m = 0; m < M; m++) { const double *o = obs_ptr(k, m); ob[3 * m] = o[0]; ob[3 * m + 1] = o[1]; ob[3 * m + 2] = o[2]; }
                }
                double sn = 0.0, cs = 0.0, dr[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
                if (NSELF) {
                    sn = TRG[k * 8]; cs = TRG[k * 8 + 1];
#pragma unroll
                    for (int a = 0; a < 3; a++) { dr[a] = TRG[k * 8 + 2 + a]; dz[a] = TRG[k * 8 + 5 + a]; }
                }
                mmpc_sched_fence();
#pragma unroll
                for (int e = 0; e < NXX; e++) hxx[e] = 0.0;
#pragma unroll
                for (int j = 0; j < NX; j++) hxx[j * (j + 1) / 2 + j] = wd[j] + reg;
                double h02 = 0.0;
                if (k < N && dyn_curv) {
                    hxx[5] += l3 * cv4 - l4 * cv3;
                    hxx[19] += dt * l3;
                    hxx[18] -= dt * l4;
                    h02 = -(-l3 * cv10 + l4 * cv9);
                }
                double hss = 2 * Sw, gss = 2 * Sw * sk, vx[6] = {0, 0, 0, 0, 0, 0};
                if (RG > 1 && M > 0) {   // circle rows: own lane's sums first, then the other lanes' in lane order
#pragma unroll
                    for (int q = 0; q < 9; q++) {
                        double c = ls.cp[q];
#pragma unroll
                        for (int g = 0; g < RG - 1; g++) c += ob[9 * g + q];
                        if (q < 3) hxx[q] += c; else if (q < 5) qx[q - 3] += c; else if (q == 5) hss += c; else if (q == 6) gss += c; else vx[q - 7] += c;
                    }
                }
#pragma unroll
                for (int m = 0; m < (RG > 1 ? 0 : M); m++) {
                    // row geometry is re-derived from x_k (cheaper than keeping it in registers)
                    const double *o = ob + 3 * m;
                    const double ddx = px - o[0], ddy = py - o[1], m2 = ddx * ddx + ddy * ddy, id = mmpc_rsqrt(m2), d = m2 * id;
                    const double g0 = -ddx * id, g1 = -ddy * id, hv = (o[2] + MMPC_BASE_R) - d - sk;
                    const double t = ls.ct[m], z = ls.cz[m], it_ = mmpc_rcp(t), w = z * it_;
                    double res = hv + t;
                    if constexpr (socm) res = io.soc[O_AR + k * NRS + m];   // corrected pass: (h + t)_soc
                    const double zh = mu * it_ + w * res;
                    hxx[0] += w * g0 * g0; hxx[1] += w * g1 * g0; hxx[2] += w * g1 * g1;
                    if (exact) { const double zi = z * id; hxx[0] -= zi * (1 - g0 * g0); hxx[1] += zi * g0 * g1; hxx[2] -= zi * (1 - g1 * g1); }
                    qx[0] += g0 * zh; qx[1] += g1 * zh;
                    hss += w; gss -= zh; vx[0] += w * g0; vx[1] += w * g1;
                }
                if (NSELF) {
#pragma unroll
                    for (int i = 0; i < NSELF; i++) {
                        double g6[6];
                        const double hv = mmpc_self_row(i, px, py, cs, sn, dr, dz, g6) - sks;
                        const double t = ls.st[i], z = ls.sz[i], it_ = mmpc_rcp(t), w = z * it_;
                        double res = hv + t;
                        if constexpr (socm) res = io.soc[O_AR + k * NRS + M + i];
                        const double zh = mu * it_ + w * res;
                        // (the terminal stage's self rows belong to s_{N-1}, quirk Q1: they do not enter this lane's s_k block)
                        const double wk = k < N ? w : 0.0, zhk = k < N ? zh : 0.0;
#pragma unroll
                        for (int a = 0; a < 6; a++) {
                            const double wa = w * g6[a];
#pragma unroll
                            for (int b = 0; b <= a; b++) hxx[mmpc_y(a) * (mmpc_y(a) + 1) / 2 + mmpc_y(b)] += wa * g6[b];
                            qx[mmpc_y(a)] += g6[a] * zh;
                            vx[a] += wk * g6[a];
                        }
                        hss += wk; gss -= zhk;
                    }
                }
                // Q1: terminal self rows -> s_{N-1}.  a = v + A^T vN, b = B^T vN, gamma = g_s - vN.c.  Lane N-1 publishes
                // (a, b, gamma, 1/h_ss); the dense rank-one blocks are applied by all lanes in the next phase, so this lane
                // stores its blocks WITHOUT the elimination of s (weight 0 below; one store sequence for every lane)
                const bool q1 = NSELF && k == N - 1;
                if (q1) {
                    const double hssN = SN[0], gssN = -(mu * SN[1] + SN[2]);
                    hss += hssN; gss += gssN;
                }
                const double ih = mmpc_rcp(hss), ihs = q1 ? 0.0 : ih;
                if (!NSELF && k == N - 1) {
                    for (int c = 0; c < NU * NX; c++) HUXL[c] = 0.0;
                    for (int c = 0; c < NUU; c++) HUUL[c] = 0.0;
                }
                {
                    constexpr int ny = NSELF ? 6 : 2;
#pragma unroll
                    for (int a = 0; a < ny; a++) {
#pragma unroll
                        for (int b = 0; b <= a; b++) hxx[mmpc_y(a) * (mmpc_y(a) + 1) / 2 + mmpc_y(b)] -= vx[a] * vx[b] * ihs;
                        qx[mmpc_y(a)] += vx[a] * gss * ihs;
                    }
                    if (k < N) {
#pragma unroll
                        for (int c = 0; c < NU; c++) QXU[k * NV + NX + c] = RB[k * NV + NX + c];
                    }
#pragma unroll
                    for (int e = 0; e < NXX; e++) HXX[k * NXX + e] = hxx[e];
#pragma unroll
                    for (int j = 0; j < NX; j++) QXU[k * NV + j] = qx[j];
                }
                if (q1) {
                    double vf[NX], a[NX], b[NU];
#pragma unroll
                    for (int j = 0; j < NX; j++) { vf[j] = 0.0; a[j] = 0.0; }
#pragma unroll
                    for (int q = 0; q < 6; q++) { vf[mmpc_y(q)] = SN[4 + q]; a[mmpc_y(q)] = vx[q]; }
                    const double *cv = CV + k * MMPC_NCV;
                    a[0] += vf[0]; a[1] += vf[1];
                    a[2] += vf[2] + cv[3] * vf[3] + cv[4] * vf[4];
                    a[3] += vf[3] + dt * vf[0] + cv[5] * vf[4];
                    a[4] += vf[4] + dt * vf[1] + cv[6] * vf[3];
                    a[5] += vf[5] + dt * vf[2] + cv[7] * vf[3] + cv[8] * vf[4];
                    b[0] = cv[9] * vf[3] + cv[10] * vf[4];
                    b[1] = dt * vf[5];
                    if (KIND == 0) { a[6] += vf[6]; a[7] += vf[7]; a[8] += vf[8]; b[2] = dt * vf[6]; b[3] = dt * vf[7]; b[4] = dt * vf[8]; }
                    double gam = gss;
#pragma unroll
                    for (int j = 0; j < NX; j++) gam -= vf[j] * CD[k * NX + j];
#pragma unroll
                    for (int j = 0; j < NX; j++) Q1V[j] = a[j];
#pragma unroll
                    for (int c = 0; c < NU; c++) Q1V[NX + c] = b[c];
                    Q1V[NV] = gam * ih; Q1V[NV + 1] = ih;
                }
                ls.hss = hss; ls.gss = gss;
#pragma unroll
                for (int a = 0; a < 6; a++) ls.vx[a] = vx[a];
                HUX02[k] = h02;
#pragma unroll
                for (int c = 0; c < NU; c++) HUUD[k * NU + c] = CST[MMPC_C_RW2 + c * NU + c] + reg;
            }
            LANES_END
            };
            if (soc_st == 1) a1_stage(MmpcTag<true>{}); else a1_stage(MmpcTag<false>{});
            if (NSELF) {
                // ---- dense rank-one blocks of stage N-1 (the elimination of s_{N-1} reaches x_N through the dynamics):
                //      Hxx -= a a^T/h, Hux = -b a^T/h, Huu = -b b^T/h, q += (a; b) gamma/h
                LANES_BEGIN
                const double ih = Q1V[NV + 1], gih = Q1V[NV];
                for (int e = lane; e < NXX + NU * NX + NUU + NV; e += MMPC_WAVE) {
                    if (e < NXX) {
                        int i = 0;
                        while ((i + 1) * (i + 2) / 2 <= e) i++;
                        const int j = e - i * (i + 1) / 2;
                        HXX[(N - 1) * NXX + e] -= Q1V[i] * Q1V[j] * ih;
                    } else if (e < NXX + NU * NX) {
                        const int c = (e - NXX) / NX, j = (e - NXX) % NX;
                        HUXL[c * NX + j] = -Q1V[NX + c] * Q1V[j] * ih;
                    } else if (e < NXX + NU * NX + NUU) {
                        const int q = e - NXX - NU * NX;
                        int c = 0;
                        while ((c + 1) * (c + 2) / 2 <= q) c++;
                        const int d2 = q - c * (c + 1) / 2;
                        HUUL[q] = -Q1V[NX + c] * Q1V[NX + d2] * ih;
                    } else {
                        const int j = e - NXX - NU * NX - NUU;
                        QXU[(N - 1) * NV + j] += Q1V[j] * gih;
                    }
                }
                LANES_END
            }
            MMPC_TS(3)
            // ---- A1 (pair lanes): barrier terms of the box rows (diagonal entries, unique owners)
            LANES_BEGIN
            auto &ls = MMPC_LS;
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                mmpc_sched_fence();
                const int idx = lane + MMPC_WAVE * p;
                if (idx < NPAIR) {
                    const int k = idx / NV, v = idx % NV;
                    const double lo = ls.b_lo[p], hi = ls.b_hi[p];
                    const bool alo = lo > -1e299, ahi = hi < 1e299;
                    // the diagonal Hessian entry this pair owns (state: HXX[k] diagonal, input: HUUD[k]); loads first
                    double *hd = v < NX ? HXX + k * NXX + v * (v + 1) / 2 + v : HUUD + (k < N ? k : 0) * NU + v - NX;
                    const double val = XU[idx], q0 = QXU[idx], h0 = *hd;
                    mmpc_sched_fence();
                    double wsum = 0.0, gsum = 0.0;
                    if (alo) { const double it_ = mmpc_rcp(mmpc_box_t(val - lo)); wsum += ls.lo_z[p] * it_; gsum -= mu * it_; }
                    if (ahi) { const double it_ = mmpc_rcp(mmpc_box_t(hi - val)); wsum += ls.hi_z[p] * it_; gsum += mu * it_; }
                    if (alo || ahi) { QXU[idx] = q0 + gsum; *hd = h0 + wsum; }
                }
            }
            LANES_END
            MMPC_TS(4)
            // ---- R0: terminal cost-to-go [P_N p_N; p_N^T .] = stage-N Hessian and gradient, in accumulator layout;
            //      operands of stage N-1: rAB = homogeneous dynamics rows 4r+g, rM = stage matrix (accumulator input)
            LANES_BEGIN
            auto &ls = MMPC_LS;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned o = ls.h_o[r];
                ls.rP[r] = ((ls.h_m >> r) & 1u) ? lds[(o & 0xffffu) + N * (int)(o >> 16)] : 0.0;
                ls.rM[r] = lds[(o & 0xffffu) + (N - 1) * (int)(o >> 16)] + lds[ls.h_l[r]];
            }
#pragma unroll
            for (int r = 0; r < NKB; r++) { const unsigned o = ls.ab_o[r]; ls.rAB[r] = lds[(o & 0xffffu) + (N - 1) * (int)(o >> 16)]; }
            LANES_END
#pragma unroll RIC_UNROLL
            for (int k = N - 1; k >= 0; k--) {
                MMPC_TS(5)
                if (ric_bad) break;   // (uniform: every lane factorises the same matrix) the pass is redone one rung down
                // R1: T = [P p; p^T .] [A B c; 0 0 1]   (symmetric: its accumulator registers are the A operand; the affine
                // part rides along as row / column NX, so the chain starts from a zero accumulator)
                MMPC_MFMA0(rT, ls.rP[0], ls.rAB[0])
                MMPC_MFMA(rT, ls.rP[1], ls.rAB[1])
                if (NKB > 2) MMPC_MFMA(rT, ls.rP[NKB > 2 ? 2 : 0], ls.rAB[NKB > 2 ? 2 : 0])
                // R2: M = [A B c; 0 0 1]^T T + stage matrix  ->  [F gx G^T; gx^T . gu^T; G gu Hh]
                MMPC_MFMA(rM, ls.rAB[0], ls.rT[0])
                MMPC_MFMA(rM, ls.rAB[1], ls.rT[1])
                if (NKB > 2) MMPC_MFMA(rM, ls.rAB[NKB > 2 ? 2 : 0], ls.rT[NKB > 2 ? 2 : 0])
                MMPC_TS(6)
                // operands of the next stage travel while this one eliminates its inputs
                LANES_BEGIN
                auto &ls = MMPC_LS;
                if (k > 0) {
#pragma unroll
                    for (int r = 0; r < NKB; r++) { const unsigned o = ls.ab_o[r]; ls.nab[r] = lds[(o & 0xffffu) + (k - 1) * (int)(o >> 16)]; }
#pragma unroll
                    for (int r = 0; r < 4; r++) { const unsigned o = ls.h_o[r]; ls.nhm[r] = lds[(o & 0xffffu) + (k - 1) * (int)(o >> 16)]; }
                }
                LANES_END_REG
                // R3: the inputs are eliminated ON the tile (block L D L^T = Schur complement).  For input a with pivot row
                // c = M[NX+1+a][.] and pivot d = c[NX+1+a] > 0,  M <- M - c c^T / d  is one rank-one MFMA whose only non-zero
                // K-slot is supplied by the lane group that already holds the row in its accumulator (row index mod 4 = lane
                // group = K-slot: no LDS exchange).  Two inputs whose rows sit in neighbouring lane groups of the same
                // accumulator register go together (one MFMA with two K-slots: the serial chain accumulator -> pivot ->
                // reciprocal -> MFMA is what a stage costs): M <- M - c0 c0^T / d0 - c1' c1'^T / d1 with c1' = c1 - (d01/d0) c0,
                // the second group fetching its partner's row entry from lane ^ 16.  What is left in rows / columns (x, 1) after the
                // last input is [P_k p_k; p_k^T .].  The normalised rows c / d are kept: u_a = -(c/d) . (dx, 1,
                // u_b>a), from which the gains are formed for all stages at once after the pass.
                constexpr bool PAIRS = F::PAIRS;                          // first input row even -> rows (2q, 2q+1) share a register
                constexpr int NLEG = F::NLEG;
#pragma unroll
                for (int leg = 0; leg < NLEG; leg++) {
                    LANES_BEGIN
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
                        // the two pivots of the pair in sequence (same arithmetic as two rank-one steps: l = d01 / d00, second
                        // pivot d1 = d11 - l d01, second row c1 - l c0), applied as the two K-slots of ONE MFMA
                        const double i0 = mmpc_rcp3(d00), l = d01 * i0, d1 = fma(-l, d01, d11);
                        if (!(d00 > 0.0 && d1 > 0.0)) ric_bad = 1;   // (NaN fails too) the pass stops at the next stage, redone one rung down
                        const double i1 = mmpc_rcp3(d1), co = MMPC_LANE_XOR16(rM[ra]);
                        const bool in1 = g == ga + 1;
                        own = g == ga || in1;
                        // (the exchanged value is used by EVERY lane, with a zero multiplier where it does not apply: written as
                        //  in1 ? fma(-l, co, c) : c the compiler may sink the cross-lane operation under the lane-dependent branch,
                        //  where it reads inactive lanes - seen with a four-input variant of this leg)
                        cb = fma(in1 ? -l : 0.0, co, c);
                        w = cb * (in1 ? i1 : i0);
                    } else {
                        const double d = MMPC_LANE_GET(rM[ra], 16 * ga + ta);
                        if (!(d > 0.0)) ric_bad = 1;
                        own = g == ga;
                        w = c * mmpc_rcp3(d);
                    }
                    ls.opa = own ? -w : 0.0;
                    ls.opb = own ? cb : 0.0;
                    // (lanes that hold no entry of the row write to their dump slot - no branch; MMPC_GK_MASK: masked instead, see there)
                    if (!(GK && MMPC_GK_MASK) || ls.kl_s[leg] != 0)
                        *(double *)((char *)KBASE + (unsigned)(ls.kl_b[leg] + MMPC_MUL24(k, ls.kl_s[leg]))) = w;
                    LANES_END_REG
                    MMPC_MFMA(rM, ls.opa, ls.opb)
                }
                MMPC_TS(7)
                LANES_BEGIN
                auto &ls = MMPC_LS;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    ls.rP[r] = ls.rM[r];
                    const unsigned o = ls.p_o[r];   // (entries that are not stored go to a dump slot: no branch)
                    lds[(o & 0xffffu) + k * (int)(o >> 16)] = ls.rM[r];
                    ls.rM[r] = ls.nhm[r];
                }
#pragma unroll
                for (int r = 0; r < NKB; r++) ls.rAB[r] = ls.nab[r];
                LANES_END
            }
            if (ric_bad) failed = 1;
            if (!failed) { if (dw > 0.0) delta_last = dw; break; }
            dw = dw == 0.0 ? (delta_last == 0.0 ? MMPC_IC_D0 : mmpc_vmax(1e-20, MMPC_IC_DN * delta_last)) : MMPC_IC_UP * dw;
            if (dw > 1e40) break;
            failed = 0;
        }
        if (failed) {
#ifdef MMPC_EMU_DEBUG
            fprintf(stderr, "riccati failed twice it %d\n", it);
#endif
            status = 2; fatal = 1; }
        if (!fatal) {
        // ---- gains of all stages from the normalised pivot rows, by back-substitution over the inputs (the last eliminated
        //      input depends on x only): K_a = -(w_a[x,1] + sum_{b>a} w_a[u_b] K_b), in place, one lane per (stage, column)
        if (GK) MMPC_GFENCE();   // the legs' stores to the gain block are read by other lanes
        LANES_BEGIN
        {
            // two (stage, column) items per lane at a time, the loads of both ahead of the arithmetic: the update is in place, so the
            // compiler cannot move the loads of one item above the stores of the one before
            constexpr int NITEM = N * (NX + 1), NTRIP = (NITEM + MMPC_WAVE - 1) / MMPC_WAVE;
            // entry (stage, input, column) of [K_k | kf_k]; in the gain block one access with a selected offset (a select between two
            // global accesses becomes a branch per access)
            auto kref = [&](int kq, int a, int jq) -> double & {
                if constexpr (GK) return gk(io.gscr, jq < NX ? GB::KK + (kq * NU + a) * NX + jq : GB::KF + kq * NU + a);
                else return jq < NX ? KK[(kq * NU + a) * NX + jq] : KF[kq * NU + a];
            };
            // (gain block in global memory: the loads of half of the lane's items ahead of their arithmetic - two exposed round trips to L2
            //  instead of one per pair of items)
            constexpr int TSTEP = GK ? (NTRIP + 1) / 2 : 2;
#pragma unroll
            for (int t0 = 0; t0 < NTRIP; t0 += TSTEP) {
                double kv[TSTEP][NU], cu[TSTEP][NPU > 0 ? NPU : 1];
                int kk[TSTEP], jj[TSTEP];
                bool ok[TSTEP];
#pragma unroll
                for (int u = 0; u < TSTEP; u++) {
                    const int i = lane + MMPC_WAVE * (t0 + u);
                    ok[u] = t0 + u < NTRIP && i < NITEM;
                    const int ii = ok[u] ? i : 0;
                    kk[u] = ii / (NX + 1); jj[u] = ii % (NX + 1);
#pragma unroll
                    for (int a = 0; a < NU; a++) kv[u][a] = kref(kk[u], a, jj[u]);
#pragma unroll
                    for (int q = 0; q < NPU; q++) cu[u][q] = gk(KU, kk[u] * NPU + q);
                }
#pragma unroll
                for (int u = 0; u < TSTEP; u++) {
#pragma unroll
                    for (int a = NU - 1; a >= 0; a--) {
                        double v = kv[u][a];
#pragma unroll
                        for (int b2 = a + 1; b2 < NU; b2++) v += cu[u][a * (2 * NU - a - 1) / 2 + (b2 - a - 1)] * kv[u][b2];
                        kv[u][a] = -v;
                    }
                }
#pragma unroll
                for (int u = 0; u < TSTEP; u++) {
                    if (ok[u]) {
#pragma unroll
                        for (int a = 0; a < NU; a++) kref(kk[u], a, jj[u]) = kv[u][a];
                    }
                }
            }
        }
        LANES_END
        if (GK) MMPC_GFENCE();
        MMPC_TS(8)
        // ---- forward roll-out: lane i < NX carries dx_k[i] in a register; a stage broadcasts the NX values through scalar
        //      registers (v_readlane), forms the input step of its row and the next dx - no LDS round trip on the chain.
        //      The operands of a stage (gain row, coefficients) do not depend on the chain and are fetched one stage ahead;
        //      dx and du are stored for the phases that follow, nothing in the loop reads them back.
        {
            double nkr[NX], nkf = 0.0, nc0 = 0.0, ncf[5];
            LANES_BEGIN
            auto &ls = MMPC_LS;
            for (int j = lane; j < NV; j += MMPC_WAVE) DXU[j] = 0.0;
            ls.fw[MMPC_FW_SLOT(0)] = 0.0;
            {   // (lanes that own no row / no input write to their dump slot: no branch in the loop)
                const int a = (int)MMPC_B(ls.f_x, 2) - 1;
                ls.fw_a[0] = lane < NX ? L.DXU + NV + lane : L.DUMP + lane;
                ls.fw_a[1] = (lane < NX && a >= 0 && lane != 4) ? L.DXU + NX + a : L.DUMP + lane;
                ls.fw_a[2] = lane < NX ? NV : 0;
                ls.fw_a[3] = lane < NX ? lane : 0;        // row of the dynamics this lane carries
                ls.fw_a[4] = a < 0 ? 0 : a;               // the input that enters this row
            }
            LANES_END
#ifndef MMPC_EMU
            // Gains in global memory (GK): lane l < GRS carries element l of a stage's (K_k, kf_k).  A ring of four stage slots in
            // LDS - over the stage-matrix extras HUXL .. HUUD, dead since the backward pass - is filled two stages ahead of the
            // chain from registers that were loaded from the gain block three stages before that; the row lanes read their gain
            // row from the ring one stage ahead, as they read it from the LDS copy of the short horizons.
            constexpr int GRS = NU * NX + NU;
            static_assert(!GK || 4 * GRS <= NU * NX + NUU + NS + NS * NU, "the gain ring must fit the stage-matrix extras");
            static_assert(!GK || FWD_UNROLL >= N, "the gain ring's registers rotate statically: the roll-out must be fully unrolled");
            double *const RING = lds + L.HUXL;
            double gpre[3] = {0.0, 0.0, 0.0};
            unsigned g_off = 0, g_st = 0;
            int r_off = 0, r_st = 0;
            auto gload = [&](int stage) -> double { return *(const double *)((const char *)io.gscr + (size_t)(g_off + (unsigned)stage * g_st)); };
            if (GK) {
                const int lane = mmpc_lane_id();
                const bool isk = lane < NU * NX, isf = !isk && lane < GRS;
                g_off = (unsigned)(isk ? GB::KK + lane : (isf ? GB::KF + lane - NU * NX : GB::DUMP + lane)) * 8u;
                g_st = isk ? (unsigned)(NU * NX) * 8u : (isf ? (unsigned)NU * 8u : 0u);
                r_off = lane < GRS ? L.HUXL + lane : L.DUMP + lane;
                r_st = lane < GRS ? GRS : 0;
                const double e0 = gload(0), e1 = N > 1 ? gload(1) : 0.0;
#pragma unroll
                for (int q = 0; q < 3; q++) if (q + 2 < N) gpre[q] = gload(q + 2);
                lds[r_off] = e0; lds[r_off + r_st] = e1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            {   // operands of stage 0
                const int lane = mmpc_lane_id();
                auto &ls = MMPC_LS;
                const int i = lane < NX ? lane : 0, a = (int)MMPC_B(ls.f_x, 2) - 1, aa = a < 0 ? 0 : a;
#pragma unroll
                for (int j = 0; j < NX; j++) nkr[j] = GK ? RING[aa * NX + j] : KK[aa * NX + j];
                nkf = GK ? RING[NU * NX + aa] : KF[aa]; nc0 = CD[i];
#pragma unroll
                for (int q = 0; q < 4; q++) ncf[q] = CV[MMPC_B(ls.f_v, q)];
                ncf[4] = CV[MMPC_B(ls.f_x, 1)];
            }
#endif
#pragma unroll FWD_UNROLL
            for (int k = 0; k < N; k++) {
                LANES_BEGIN
                auto &ls = MMPC_LS;
                const int i = ls.fw_a[3], aa = ls.fw_a[4];
                const int fw_dx = ls.fw_a[0], fw_du = ls.fw_a[1], fw_st = ls.fw_a[2];
                double kr[NX], cf[5], kf0, c0;
#ifdef MMPC_EMU
                {
                    const double *cv = CV + k * MMPC_NCV;
                    for (int j = 0; j < NX; j++) kr[j] = KK[(k * NU + aa) * NX + j];
                    kf0 = KF[k * NU + aa]; c0 = CD[k * NX + i];
                    for (int q = 0; q < 4; q++) cf[q] = cv[MMPC_B(ls.f_v, q)];
                    cf[4] = cv[MMPC_B(ls.f_x, 1)];
                }
#else
#pragma unroll
                for (int j = 0; j < NX; j++) kr[j] = nkr[j];
                kf0 = nkf; c0 = nc0;
#pragma unroll
                for (int q = 0; q < 5; q++) cf[q] = ncf[q];
                if (k + 1 < N) {
                    const double *cv = CV + (k + 1) * MMPC_NCV;
#pragma unroll
                    for (int j = 0; j < NX; j++) nkr[j] = GK ? RING[((k + 1) & 3) * GRS + aa * NX + j] : KK[((k + 1) * NU + aa) * NX + j];
                    nkf = GK ? RING[((k + 1) & 3) * GRS + NU * NX + aa] : KF[(k + 1) * NU + aa]; nc0 = CD[(k + 1) * NX + i];
#pragma unroll
                    for (int q = 0; q < 4; q++) ncf[q] = cv[MMPC_B(ls.f_v, q)];
                    ncf[4] = cv[MMPC_B(ls.f_x, 1)];
                }
                if (GK) {
                    if (k + 2 < N) lds[r_off + ((k + 2) & 3) * r_st] = gpre[k % 3];
                    if (k + 5 < N) gpre[k % 3] = gload(k + 5);
                }
#endif
                double dx[NX];
#pragma unroll
                for (int j = 0; j < NX; j++) dx[j] = MMPC_LANE_GET(fw[MMPC_FW_SLOT(k)], j);
                // du_a = kf_a + K_a dx in three partial sums; the terms of dx+ that do not need du meanwhile
                double d0 = kf0, d1 = 0.0, d2 = 0.0;
#pragma unroll
                for (int j = 0; j < NX; j += 3) {
                    d0 += kr[j] * dx[j];
                    if (j + 1 < NX) d1 += kr[j + 1] * dx[j + 1];
                    if (j + 2 < NX) d2 += kr[j + 2] * dx[j + 2];
                }
                double v = c0 + ls.fw[MMPC_FW_SLOT(k)];
                double w = cf[0] * dx[2];
                v += cf[1] * dx[3];
                w += cf[2] * dx[4];
                v += cf[3] * dx[5];
                const double du = d0 + (d1 + d2);
                v = (v + w) + cf[4] * du;
                ls.fw[MMPC_FW_SLOT(k + 1)] = v;
                // (lanes that own no row / no input write to their dump slot: no branch)
                lds[fw_du + k * fw_st] = du;
                lds[fw_dx + k * fw_st] = v;
                LANES_END_REG
            }
            LANES_BEGIN
            LANES_END
        }
        MMPC_TS(9)
        // ---- D1: multiplier step and slack-variable step (stage lanes)
        LANES_BEGIN
        auto &ls = MMPC_LS;
        if (lane < NS) {
            const int k = lane;
            // lam_k + dlam_k = -(P_k dx_k + p_k): every load ahead of the arithmetic, the stores at the end (a store inside the
            // row loop makes the compiler wait for each load: it cannot tell DLAM from HXX)
            double dx[NX], pk[NXX], y[NX];
#pragma unroll
            for (int j = 0; j < NX; j++) { dx[j] = DXU[k * NV + j]; y[j] = QXU[k * NV + j] + LAM[k * NX + j]; }
#pragma unroll
            for (int e = 0; e < NXX; e++) pk[e] = HXX[k * NXX + e];
            mmpc_sched_fence();
#pragma unroll
            for (int i = 0; i < NX; i++) {
#pragma unroll
                for (int j = 0; j < NX; j++) y[i] += pk[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i] * dx[j];
            }
#pragma unroll
            for (int i = 0; i < NX; i++) DLAM[k * NX + i] = -y[i];
            double vdx = 0.0;
            constexpr int ny = NSELF ? 6 : 2;
#pragma unroll
            for (int a = 0; a < ny; a++) vdx += ls.vx[a] * dx[mmpc_y(a)];
            if (k == N - 1 && NSELF) {
#pragma unroll
                for (int a = 0; a < 6; a++) vdx += SN[4 + a] * DXU[N * NV + mmpc_y(a)];
            }
            DS[k] = -(ls.gss - vdx) * mmpc_rcp(ls.hss);
            if (k == N) { for (int c = 0; c < NU; c++) DXU[N * NV + NX + c] = 0.0; }
        }
        LANES_END
        }   // !fatal
        }   // soc_st != 3
        if (fatal) break;
        MMPC_TS(10)
        // ---- D2: row steps, fraction-to-boundary, directional derivative
        const double tau = mmpc_vmax(0.99, 1.0 - mu);
        auto d2_rows = [&](auto TAG) {
        constexpr bool socm = decltype(TAG)::value; (void)socm;
        LANES_BEGIN
        auto &ls = MMPC_LS;
        // fraction to the boundary without divisions or branches: alpha = min(1, tau / max_i(-dt_i / t_i)) (1/t_i is at hand),
        // likewise for the multipliers with 1/z_i
        double rp = 0.0, rd = 0.0, dphi = 0.0;
        if (RG > 1 && M > 0) {   // circle rows where RG lanes share a stage
            MMPC_ROW_LANE
            if (rlane) {
                const double dx0 = DXU[rk * NV], dx1 = DXU[rk * NV + 1], dsk = DS[rk];
                const double px = XU[rk * NV], py = XU[rk * NV + 1], sk = S[rk];
                double ob[MCR * 3];
#pragma unroll
                for (int r = 0; r < MCR; r++) {
                    const double *o = obs_ptr(rk, rs + RG * r < M ? rs + RG * r : 0);
                    ob[3 * r] = o[0]; ob[3 * r + 1] = o[1]; ob[3 * r + 2] = o[2];
                }
#pragma unroll
                for (int r = 0; r < MCR; r++) {
                    if (rs + RG * r < M) {
                        const double ddx = px - ob[3 * r], ddy = py - ob[3 * r + 1], m2 = ddx * ddx + ddy * ddy, id = mmpc_rsqrt(m2), d = m2 * id;
                        const double hv = (ob[3 * r + 2] + MMPC_BASE_R) - d - sk;
                        const double t = ls.ct[r], z = ls.cz[r];
                        const double jd = -(ddx * dx0 + ddy * dx1) * id - dsk;
                        double res = hv + t;
                        if constexpr (socm) res = io.soc[O_AR + rk * NRS + rs + RG * r];
                        const double dtv = -res - jd, it_ = mmpc_rcp(t), dzv = mu * it_ - z - z * it_ * dtv;
                        ls.cdt[r] = dtv;
                        rp = mmpc_vmax(rp, -dtv * it_);
                        rd = mmpc_vmax(rd, -dzv * mmpc_rcp(z));
                        dphi -= mu * dtv * it_;
                    }
                }
            }
        }
        if (lane < NS) {
            const int k = lane;
            const double *dx = DXU + k * NV;
            const double dsk = DS[k], dsks = DS[slack_idx(k)];
            const double px = XU[k * NV], py = XU[k * NV + 1], sk = S[k];
            double ob[(MC > 0 ? MC : 1) * 3];
#pragma unroll
            for (int m = 0; m < (RG > 1 ? 0 : M); m++) { const double *o = obs_ptr(k, m); ob[3 * m] = o[0]; ob[3 * m + 1] = o[1]; ob[3 * m + 2] = o[2]; }
#pragma unroll
            for (int m = 0; m < (RG > 1 ? 0 : M); m++) {
                const double ddx = px - ob[3 * m], ddy = py - ob[3 * m + 1], m2 = ddx * ddx + ddy * ddy, id = mmpc_rsqrt(m2), d = m2 * id;
                const double hv = (ob[3 * m + 2] + MMPC_BASE_R) - d - sk;
                const double t = ls.ct[m], z = ls.cz[m];
                const double jd = -(ddx * dx[0] + ddy * dx[1]) * id - dsk;
                double res = hv + t;
                if constexpr (socm) res = io.soc[O_AR + k * NRS + m];
                const double dtv = -res - jd, it_ = mmpc_rcp(t), dzv = mu * it_ - z - z * it_ * dtv;
                ls.cdt[m] = dtv;
                rp = mmpc_vmax(rp, -dtv * it_);
                rd = mmpc_vmax(rd, -dzv * mmpc_rcp(z));
                dphi -= mu * dtv * it_;
            }
            double sn = 0.0, cs = 0.0, dr[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
            if (NSELF) {
                sn = TRG[k * 8]; cs = TRG[k * 8 + 1];
#pragma unroll
                for (int a = 0; a < 3; a++) { dr[a] = TRG[k * 8 + 2 + a]; dz[a] = TRG[k * 8 + 5 + a]; }
            }
            const double sks = S[slack_idx(k)];
#pragma unroll
            for (int i = 0; i < NSELF; i++) {
                double g6[6];
                const double hv = mmpc_self_row(i, px, py, cs, sn, dr, dz, g6) - sks;
                const double t = ls.st[i], z = ls.sz[i];
                double jd = -dsks;
#pragma unroll
                for (int a = 0; a < 6; a++) jd += g6[a] * dx[mmpc_y(a)];
                double res = hv + t;
                if constexpr (socm) res = io.soc[O_AR + k * NRS + M + i];
                const double dtv = -res - jd, it_ = mmpc_rcp(t), dzv = mu * it_ - z - z * it_ * dtv;
                ls.sdt[i] = dtv;
                rp = mmpc_vmax(rp, -dtv * it_);
                rd = mmpc_vmax(rd, -dzv * mmpc_rcp(z));
                dphi -= mu * dtv * it_;
            }
            dphi += 2 * Sw * S[k] * dsk;
        }
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            mmpc_sched_fence();
            const int idx = lane + MMPC_WAVE * p;
            if (idx < NPAIR) {
                const int k = idx / NV, v = idx % NV;
                const double lo = ls.b_lo[p], hi = ls.b_hi[p];
                const bool alo = lo > -1e299, ahi = hi < 1e299;
                const double val = XU[idx], dv = DXU[idx];
                dphi += RB[idx] * dv;   // RB = plain cost gradient of this variable (kept by the evaluation)
                if (alo) {
                    const double t = mmpc_box_t(val - lo), z = ls.lo_z[p], dtv = dv, it_ = mmpc_rcp(t), dzv = mu * it_ - z - z * it_ * dtv;
                    rp = mmpc_vmax(rp, -dtv * it_);
                    rd = mmpc_vmax(rd, -dzv * mmpc_rcp(z));
                    dphi -= mu * dtv * it_;
                }
                if (ahi) {
                    const double t = mmpc_box_t(hi - val), z = ls.hi_z[p], dtv = -dv, it_ = mmpc_rcp(t), dzv = mu * it_ - z - z * it_ * dtv;
                    rp = mmpc_vmax(rp, -dtv * it_);
                    rd = mmpc_vmax(rd, -dzv * mmpc_rcp(z));
                    dphi -= mu * dtv * it_;
                }
            }
        }
        MMPC_WR(0) = rp; MMPC_WR(1) = rd; MMPC_WR(2) = dphi;
        LANES_END
        };
        if (soc_st == 1) d2_rows(MmpcTag<true>{}); else d2_rows(MmpcTag<false>{});
        {
            const double rp_ = MMPC_RED_MAX(0), rd_ = MMPC_RED_MAX(1), ap_n = rp_ > tau ? tau * mmpc_rcp(rp_) : 1.0;
            if (soc_st == 1) {
                // corrected direction: its primal step from x_k (the multipliers have taken their step; the acceptance test
                // uses the length and the slope of the step it corrects)
                a_soc = ap_n;
                apply_step(a_soc, false);
                soc_st = 2; in_ls = 1;
                continue;
            }
            ap = ap_n; ad = rd_ > tau ? tau * mmpc_rcp(rd_) : 1.0; dphi = MMPC_RED_SUM(2);
            if (soc_st == 3) {
                // the correction failed: the row steps of the uncorrected direction are back in the registers, the line search
                // goes on with its second trial
                alpha = 0.5 * ap; lsi = 1; lspass = 0;
                apply_step(alpha, false);
                soc_st = 0; in_ls = 1;
                continue;
            }
        }

        MMPC_TS(11)
        if (!filt_init) { nfilt = 0; th_max = 1e4 * mmpc_vmax(1.0, th0); th_min = 1e-4 * mmpc_vmax(1.0, th0); filt_init = 1; }
        // ---- first trial of the line search: multipliers with alpha_d, primal variables / slacks with alpha = alpha_p
        alpha = ap; lspass = 0; lsi = 0;
        apply_step(alpha, true);
        in_ls = 1;
        it++;
    }

    MMPC_TS(13)
    MMPC_TEND()
    // ------------------------------------------------------------------ results
    LANES_BEGIN
    double f = 0.0;
    for (int idx = lane; idx < NPAIR; idx += MMPC_WAVE) {
        const int k = idx / NV, v = idx % NV;
        const double val = XU[idx];
        const double ref = ref_at(idx, k, v);
        double e = val - ref;
        if (KIND == 1 && v == 2) e = mmpc_angle_diff(val, ref);
        f += 0.5 * w_diag(k, v) * e * e;
        if (v >= NX && k < N) { const double e2 = val - ulast_at(k, v - NX); f += 0.5 * CST[MMPC_C_WW + v - NX] * e2 * e2; }
        if (v < NX) io.X[k * NX + v] = val;
        else if (k < N) io.U[k * NU + v - NX] = val;
    }
    for (int i = lane; i < NS; i += MMPC_WAVE) { io.s[i] = S[i]; f += Sw * S[i] * S[i]; }
    MMPC_WR(0) = f;
    LANES_END
    const double cost = MMPC_RED_SUM(0);
    LANES_BEGIN
    if (lane == 0) { *io.status = status; *io.iters = it; *io.cost = cost; *io.err = E0; }
    LANES_END
}
