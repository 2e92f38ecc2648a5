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
// stamps inside the stage loop of the backward pass (three per stage: reading the counter waits for the scalar cache and drains the
// LDS queue, ~5 k cycles per iteration in all); -DMMPC_STAMP_COARSE leaves them out - the pass is then one segment, booked under stamp 8
#ifdef MMPC_STAMP_COARSE
#define MMPC_TSF(i)
#else
#define MMPC_TSF(i) MMPC_TS(i)
#endif
#else
#define MMPC_T0()
#define MMPC_TS(i)
#define MMPC_TSF(i)
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

#ifndef MMPC_SAFEGUARD_LAZY
#define MMPC_SAFEGUARD_LAZY 1   // multiplier safeguard of the first trial: range check per row, exact clamp only if some row needs it (see apply_step; A/B switch)
#endif
// scheduler fences of this file by site: bit i of MMPC_FENCE_MASK keeps fence i (0 init, 1 trial move pairs, 2 E1 stage loads, 3 / 4 E1 pair
// pass / loads, 5 / 6 A1 pair pass / loads, 7 leg before the product, 8 D1 loads, 9 A1 stage loads, 10 D2 pairs).  A lone wave issues a
// dependent instruction every ~8 cycles and an independent one every ~4: fences that keep the unrolled bodies of a lane-parallel loop
// apart (rounds 1-3: fewer live registers) cost more in issue bubbles than the register moves they save.
#ifndef MMPC_FENCE_MASK
#define MMPC_FENCE_MASK 0x000
#endif
#define MMPC_SFENCE(i) { if ((MMPC_FENCE_MASK >> (i)) & 1) mmpc_sched_fence(); }
#ifndef MMPC_TRGS
#define MMPC_TRGS 9      // stride of the per-stage trig cache (8 words used): with 8 the stage lanes' words sit 16 dwords apart - two LDS banks for 21 lanes
#endif
#ifndef MMPC_LEG_DET
#define MMPC_LEG_DET 1    // pair legs: the second pivot's reciprocal from the 2 x 2 determinant, beside the first's (A/B switch)
#endif
#ifndef MMPC_D2_ONE_RCP
#define MMPC_D2_ONE_RCP 1   // row steps D2: one reciprocal per row (see mmpc_fast_d2.inc; A/B switch)
#endif
#ifndef MMPC_FWD_DPP
#define MMPC_FWD_DPP 1   // forward roll-out: dx_k by v_mov_b64_dpp row_newbcast instead of v_readlane pairs (A/B switch)
#endif
#ifndef MMPC_PADMAP
#define MMPC_PADMAP 1    // padded pair map (see MmpcFastDims::PADMAP); 0: the plain map everywhere (A/B switch)
#endif
#ifndef MMPC_SLIM_NMIN
#define MMPC_SLIM_NMIN 21  // horizons from here on: "slim" LDS layout (see mmpc_fast_layout) and circle rows spread over lanes
#endif
#ifndef MMPC_RG_NMIN
#define MMPC_RG_NMIN MMPC_SLIM_NMIN   // horizons from here on spread the circle rows of a stage over RG lanes (A/B switch)
#endif
#ifndef MMPC_RG_MAX
#define MMPC_RG_MAX 3
#endif
template <int KIND, int N>
struct MmpcFastDims {
    typedef MmpcDims<KIND> D;
    static constexpr int NX = D::NX, NU = D::NU, NV = D::NV, NXX = D::NXX, NUU = D::NUU, NSELF = D::NSELF;
    static constexpr int NS = N + 1;
    static constexpr int NPAIR = NS * NV;
    // Pair lanes: which (stage, variable) pair lane l works on in pass p.  Plain map: pair l + 64 p (k = idx / NV, v = idx % NV - a
    // division by 14 and the addresses derived from it, ~25 instructions per pass in each of the four pair-lane phases of an
    // iteration).  Padded map (where it needs no extra pass): a 16-lane row of the wave owns one stage per pass, row 4 p + (l >> 4),
    // column l & 15 < NV is the variable; the pairs of the terminal stage (states only) sit in column NV of the first NX rows.  Every
    // address is then base(l) + p * stride(l), both formed once per phase.
    static constexpr bool PADMAP = MMPC_PADMAP && NV < 16 && NX <= N && (N * 16 + MMPC_WAVE - 1) / MMPC_WAVE <= (NPAIR + MMPC_WAVE - 1) / MMPC_WAVE;
    // stride of the per-stage trig cache (8 words used; the long horizons have no LDS to spare for the padding)
    static constexpr int TRGS = N >= MMPC_SLIM_NMIN ? 8 : MMPC_TRGS;
    static constexpr int NPASS = PADMAP ? (N * 16 + MMPC_WAVE - 1) / MMPC_WAVE : (NPAIR + MMPC_WAVE - 1) / MMPC_WAVE;
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
    static constexpr int RG = ((N >= MMPC_RG_NMIN || KIND == 1) && 2 * NS <= MMPC_WAVE) ? (MMPC_WAVE / NS < MMPC_RG_MAX ? MMPC_WAVE / NS : MMPC_RG_MAX) : 1;
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
    MMPC_CARVE(CST, MMPC_C_SIZE) MMPC_CARVE(CV, F::NS * MMPC_NCV) MMPC_CARVE(CD, F::NS * F::NX) MMPC_CARVE(TRG, F::NS * F::TRGS)
    MMPC_CARVE(HXX, F::NS * F::NXX) MMPC_CARVE(QXU, F::NS * F::NV) MMPC_CARVE(HUXL, F::NU * F::NX)
    MMPC_CARVE(HUUL, F::NUU) MMPC_CARVE(HUX02, F::NS) MMPC_CARVE(HUUD, F::NS * F::NU) MMPC_CARVE(SN, 16)
    // (gains in LDS: one array [stage][input][NX + 1], the feed-forward kf as column NX of the gain row - the back-substitution and the
    //  roll-out then address a row of NX + 1 words; in the gain block of the long horizons K and kf stay apart, see MmpcGainBlock)
    MMPC_CARVE(KK, GK ? 0 : N * F::NU * (F::NX + 1)) MMPC_CARVE(KF, 0) MMPC_CARVE(DXU, F::NS * F::NV) MMPC_CARVE(DS, F::NS)
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
    double mg[4], pv0, pv1;                          // elimination legs: 0 / 1 masks of the four lane groups; the pivots of the leg under way
    double nab[F::NKB], nhm[4];                      // next stage's dynamics rows and stage-matrix entries (loaded one stage ahead)
    // forward roll-out, row `lane` of [A B] (base.py:19-26): dx+[i] = dx[i] + sum_{j=2..5} C_j dx[j] + C_u du_a + c[i];
    // f_v: coefficient ids (into CV[k]) of C_2..C_5, f_x: id of C_u << 8 | (a + 1) << 16 (a: the input of this row, -1 none)
    unsigned f_v, f_x;
    unsigned q1d[2];                                 // items of the stage-(N-1) rank-one blocks (quirk Q1) this lane forms: Q1V index | Q1V index << 5 |
                                                     // accumulate << 10 | gradient item << 11 | LDS word offset of the target << 12
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
MMPC_WAVE_RED4(mmpc_wave_sum4, mmpc_op_add)
MMPC_WAVE_RED4(mmpc_wave_max4, mmpc_vmax)
#define MMPC_RED4_SUM(i0, i1, i2, i3, out) mmpc_wave_sum4(wr_one[i0], wr_one[i1], wr_one[i2], wr_one[i3], out)
#define MMPC_RED4_MAX(i0, i1, i2, i3, out) mmpc_wave_max4(wr_one[i0], wr_one[i1], wr_one[i2], wr_one[i3], out)
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
#if !MMPC_RED_DPP
#define MMPC_RED4_SUM(i0, i1, i2, i3, out) { (out)[0] = MMPC_RED_SUM(i0); (out)[1] = MMPC_RED_SUM(i1); (out)[2] = MMPC_RED_SUM(i2); (out)[3] = MMPC_RED_SUM(i3); }
#define MMPC_RED4_MAX(i0, i1, i2, i3, out) { (out)[0] = MMPC_RED_MAX(i0); (out)[1] = MMPC_RED_MAX(i1); (out)[2] = MMPC_RED_MAX(i2); (out)[3] = MMPC_RED_MAX(i3); }
#endif
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
// the four-at-a-time reductions of the device (MMPC_WAVE_RED4): lane l with l + 32, the result with its neighbour 16 lanes on, then the
// in-row butterfly of mmpc_emu_red over 16 lanes
static inline double mmpc_emu_red4(double (*wr)[9], int i, int op) {
    if (!MMPC_RED_DPP) return mmpc_emu_red(wr, i, op);
    double v[16], w[16];
    for (int l = 0; l < 16; l++) {
        const double a0 = wr[l][i], a1 = wr[l + 16][i], a2 = wr[l + 32][i], a3 = wr[l + 48][i];
        const double s0 = op == 0 ? a0 + a2 : (a0 > a2 ? a0 : a2), s1 = op == 0 ? a1 + a3 : (a1 > a3 ? a1 : a3);
        v[l] = op == 0 ? s0 + s1 : (s0 > s1 ? s0 : s1);
    }
    for (int q = 0; q < 4; q++) {
        const int o = 1 << q;
        for (int l = 0; l < 16; l++) {
            const int p = o == 4 ? ((l & ~7) | (7 - (l & 7))) : (o == 8 ? (15 - l) : (l ^ o));
            w[l] = op == 0 ? v[l] + v[p] : (v[l] > v[p] ? v[l] : v[p]);
        }
        for (int l = 0; l < 16; l++) v[l] = w[l];
    }
    return v[0];
}
#define MMPC_RED4_SUM(i0, i1, i2, i3, out) { (out)[0] = mmpc_emu_red4(wr_all, i0, 0); (out)[1] = mmpc_emu_red4(wr_all, i1, 0); (out)[2] = mmpc_emu_red4(wr_all, i2, 0); (out)[3] = mmpc_emu_red4(wr_all, i3, 0); }
#define MMPC_RED4_MAX(i0, i1, i2, i3, out) { (out)[0] = mmpc_emu_red4(wr_all, i0, 1); (out)[1] = mmpc_emu_red4(wr_all, i1, 1); (out)[2] = mmpc_emu_red4(wr_all, i2, 1); (out)[3] = mmpc_emu_red4(wr_all, i3, 1); }
#endif

// pair lanes (MmpcFastDims::PADMAP): per phase, the lane's base index and per-pass stride; per pass, the pair's index into XU / DXU / RB /
// QXU (`idx`), whether the slot holds a pair (`pok`), its stage and variable (`k`, `v`)
#define MMPC_PAIR_SETUP                                                                                                         \
    const int pg_ = lane >> 4, pc_ = lane & 15; const bool pt_ = F::PADMAP && pc_ == NV;                                         \
    const int pb_ = F::PADMAP ? (pt_ ? N * NV + pg_ : pg_ * NV + pc_) : lane, ps_ = F::PADMAP ? (pt_ ? 4 : 4 * NV) : MMPC_WAVE;   \
    (void)pg_; (void)pc_; (void)pt_;
#define MMPC_PAIR(p)                                                                                                            \
    const int idx = pb_ + (p) * ps_;                                                                                            \
    const bool pok = F::PADMAP ? (pt_ ? 4 * (p) + pg_ < NX : (pc_ < NV && 4 * (p) + pg_ < N)) : idx < NPAIR;
#define MMPC_PAIR_KV(p)                                                                                                         \
    const int k = F::PADMAP ? (pt_ ? N : 4 * (p) + pg_) : idx / NV, v = F::PADMAP ? (pt_ ? 4 * (p) + pg_ : pc_) : idx % NV;

// compile-time switch handed to the generic lambdas of the assembly / row-step phases (corrected pass of the second-order correction or not)
template <bool B> struct MmpcTag { static constexpr bool value = B; };

// product accumulator for sum(log t): the slacks themselves are multiplied and ONE logarithm is taken per lane and phase.  A lane
// multiplies at most ten box slacks (>= 1e-15 by mmpc_box_t) or nine row slacks between init() and value(): the product stays a normal
// number down to slacks of 1e-30, and scaling by powers of two being exact it has the mantissa the product of the mantissas had
// (rounds 1-3 split every factor with frexp: four instructions per row instead of one).  `ex` carries the exponent handed over
// between the two evaluation phases of the long horizons.
struct MmpcLogAcc {
    double mant; int ex;
    MMPC_DEV void init() { mant = 1.0; ex = 0; }
    MMPC_DEV void mul(double t) { mant *= t; }
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
    double *const KK = GK ? io.gscr + GB::KK : lds + L.KK, *const KF = GK ? io.gscr + GB::KF : lds + L.KK + NX, *const KU = GK ? io.gscr + GB::KU : lds + L.KU;
    // stride of a gain row / of its feed-forward entry between inputs (LDS: rows of NX + 1 words with kf as the last one - KF points at it)
    constexpr int KS = GK ? NX : NX + 1, KFS = GK ? 1 : NX + 1;
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
                    if (!GK && j <= NX) { off = (unsigned)(L.KK + a * KS + j); stride = NU * KS; }
                    else if (j < NX) { off = (unsigned)(O_KK + a * NX + j); stride = NU * NX; }
                    else if (j == NX) { off = (unsigned)(O_KF + a); stride = NU; }
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
            // items e = lane + 64 t of the dense blocks the elimination of s_{N-1} adds to stage N-1 (see the backward pass):
            // Hxx -= a a^T / h (packed lower triangle), Hux = -b a^T / h, Huu = -b b^T / h, q += (a; b) gamma / h with (a; b) = Q1V[0 .. NV),
            // gamma / h = Q1V[NV]; slots without an item multiply two zeros of the constant block into the lane's dump slot
            static_assert(NXX + NU * NX + NUU + NV <= 2 * MMPC_WAVE && NV + 2 <= 32, "stage-(N-1) block items: two per lane");
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int e = lane + MMPC_WAVE * t;
                unsigned ia = 0, ib = 0, acc = 0, grad = 0, dst = (unsigned)(L.DUMP + lane);
                if (e < NXX) {
                    int i = 0;
                    while ((i + 1) * (i + 2) / 2 <= e) i++;
                    ia = (unsigned)i; ib = (unsigned)(e - i * (i + 1) / 2); acc = 1; dst = (unsigned)(L.HXX + (N - 1) * NXX + e);
                } else if (e < NXX + NU * NX) {
                    const int c = (e - NXX) / NX, j = (e - NXX) % NX;
                    ia = (unsigned)(NX + c); ib = (unsigned)j; dst = (unsigned)(L.HUXL + c * NX + j);
                } else if (e < NXX + NU * NX + NUU) {
                    const int q = e - NXX - NU * NX;
                    int c = 0;
                    while ((c + 1) * (c + 2) / 2 <= q) c++;
                    ia = (unsigned)(NX + c); ib = (unsigned)(NX + q - c * (c + 1) / 2); dst = (unsigned)(L.HUUL + q);
                } else if (e < NXX + NU * NX + NUU + NV) {
                    const int j = e - NXX - NU * NX - NUU;
                    ia = (unsigned)j; ib = (unsigned)NV; acc = 1; grad = 1; dst = (unsigned)(L.QXU + (N - 1) * NV + j);
                }   // (else no item: -(Q1V[0]^2) / h into the lane's dump slot)
                ls.q1d[t] = ia | (ib << 5) | (acc << 10) | (grad << 11) | (dst << 12);
            }
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
            alo = ex && mmpc_bound_active(lo); ahi = ex && mmpc_bound_active(hi);
        } else {
            const int a = v - NX;
            const double ul = ulast_at(k, a);
            lo = mmpc_vmax(CST[MMPC_C_ULIM + a], ul + CST[MMPC_C_DULIM + a]);
            hi = mmpc_vmin(CST[MMPC_C_ULIM + 5 + a], ul + CST[MMPC_C_DULIM + 5 + a]);
            const bool ex = k < N;
            alo = ex && mmpc_bound_active(lo); ahi = ex && mmpc_bound_active(hi);
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
    MMPC_PAIR_SETUP
#pragma unroll
    for (int p = 0; p < NPASS; p++) {
        MMPC_SFENCE(0)
        MMPC_PAIR(p)
        ls.lo_z[p] = 0.0; ls.hi_z[p] = 0.0; ls.b_lo[p] = -1e300; ls.b_hi[p] = 1e300;
        if (pok) {
            MMPC_PAIR_KV(p)
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
    double prox_r = 0.0, delta_r = 0.0, dprev_r = 0.0;
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
        it = (int)st_scal[4]; nfilt = (int)st_scal[5]; filt_init = (int)st_scal[6]; nsmall_r = (int)st_scal[7]; delta_r = st_scal[8]; dprev_r = st_scal[9];
    }
    const int it_start = CONT ? it : 0;
    // results of the evaluation of the current point (iterate or line-search trial)
    double err_d = 0.0, err_p = 0.0, tzmax = 0.0, tzmin = 0.0, zsum = 0.0, cost_c = 0.0, th_c = 0.0, sumlog = 0.0;
    // line-search state: the evaluation of a trial point IS the evaluation the next iteration starts from (98.6 % of the
    // first trials are accepted), so the loop below evaluates once per trial and never a second time for the accepted one
    int in_ls = 0, lspass = 0, lsi = 0, nsmall = nsmall_r;
    double prox = prox_r;   // proximal term for crawling iterations (mmpc_prox_update)
    double delta_last = delta_r;   // last successful inertia correction
    double delta_prev = dprev_r;   // the correction of the previous iteration's matrix (0: none)
    double alpha = 0.0, ap = 1.0, ad = 1.0, dphi = 0.0, phi0 = 0.0, th0 = 0.0;

    // ---- move to a trial point: the primal variables, the equality multipliers and the slacks of the nonlinear rows by
    //      d_alpha times the direction (signed: a rejected trial is moved back by the difference); with `first` also the
    //      row multipliers by alpha_d (they do not depend on alpha; evaluated with the slacks of the point the step starts at)
    auto apply_step = [&](double d_alpha, bool first) {
        // Multiplier safeguard of the first trial (mmpc_z_safeguard: z clamped to [mu / (kappa t), kappa mu / t], kappa = 1e10), lazily:
        // the clamp needs 1 / t_new per row and changes nothing unless t z has left [mu / kappa, kappa mu] - which no solve of the
        // bench batches ever does.  The rows only track the range of t z (a multiply, a min, a max); if any row of the wave comes
        // within a factor 2 of either end, a second phase clamps every row exactly, from the same (z, t, mu) - the result is the one
        // of the eager clamp in every case.
        double pz_lo = 1e300, pz_hi = 0.0;
        auto zsafe = [&](double zn, double tn) -> double {
            if (!MMPC_SAFEGUARD_LAZY) return mmpc_z_safeguard_fast(zn, tn, mu);
            const double pz = zn * tn;
            pz_lo = mmpc_vmin(pz_lo, pz); pz_hi = mmpc_vmax(pz_hi, pz);
            return zn;
        };
        LANES_BEGIN
        auto &ls = MMPC_LS;
        {
            MMPC_ROW_LANE
            if (rlane) {
#pragma unroll
                for (int r = 0; r < MCR; r++) {
                    if (RG == 1 || rs + RG * r < M) {
                        const double t = ls.ct[r], z = ls.cz[r], dtv = ls.cdt[r], tn = t + d_alpha * dtv;
                        if (first) { const double it_ = mmpc_rcp(t); ls.cz[r] = zsafe(z + ad * (mu * it_ - z - z * it_ * dtv), tn); }
                        ls.ct[r] = tn;
                    }
                }
            }
        }
        if (lane < NS) {
#pragma unroll
            for (int i = 0; i < NSELF; i++) {
                const double t = ls.st[i], z = ls.sz[i], dtv = ls.sdt[i], tn = t + d_alpha * dtv;
                if (first) { const double it_ = mmpc_rcp(t); ls.sz[i] = zsafe(z + ad * (mu * it_ - z - z * it_ * dtv), tn); }
                ls.st[i] = tn;
            }
        }
        // (same phase: every lane touches only its own registers and its own words of XU / LAM / S)
        MMPC_PAIR_SETUP
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            MMPC_SFENCE(1)
            MMPC_PAIR(p)
            if (pok) {
                const double val = XU[idx], dv = DXU[idx], vn = val + d_alpha * dv;
                if (first) {
                    const double lo = ls.b_lo[p], hi = ls.b_hi[p];
                    const bool alo = lo > -1e299, ahi = hi < 1e299;
                    if (alo) { const double z = ls.lo_z[p], it_ = mmpc_rcp(mmpc_box_t(val - lo)); ls.lo_z[p] = zsafe(z + ad * (mu * it_ - z - z * it_ * dv), mmpc_box_t(vn - lo)); }
                    if (ahi) { const double z = ls.hi_z[p], it_ = mmpc_rcp(mmpc_box_t(hi - val)); ls.hi_z[p] = zsafe(z + ad * (mu * it_ - z + z * it_ * dv), mmpc_box_t(hi - vn)); }
                }
                if (idx >= NX) XU[idx] = vn;   // x_0 is data
            }
        }
        for (int i = lane; i < NS * NX; i += MMPC_WAVE) if (i >= NX) LAM[i] += d_alpha * DLAM[i];
        for (int i = lane; i < NS; i += MMPC_WAVE) S[i] += d_alpha * DS[i];
        LANES_END
        // (-DMMPC_SAFEGUARD_FORCE: the exact clamp after every first trial - the tests' way into the branch no solve takes by itself)
#ifdef MMPC_SAFEGUARD_FORCE
        constexpr bool zforce = true;
#else
        constexpr bool zforce = false;
#endif
        if (MMPC_SAFEGUARD_LAZY && first && MMPC_WAVE_ANY(zforce || !(pz_lo >= 2.0 * mu * (1.0 / MMPC_KAPPA_SIGMA) && pz_hi <= 0.5 * MMPC_KAPPA_SIGMA * mu))) {
            // (rare) the exact clamp of every row, at the slacks of the trial point
            LANES_BEGIN
            auto &ls = MMPC_LS;
            {
                MMPC_ROW_LANE
                if (rlane) {
#pragma unroll
                    for (int r = 0; r < MCR; r++) if (RG == 1 || rs + RG * r < M) ls.cz[r] = mmpc_z_safeguard_fast(ls.cz[r], ls.ct[r], mu);
                }
            }
            if (lane < NS) {
#pragma unroll
                for (int i = 0; i < NSELF; i++) ls.sz[i] = mmpc_z_safeguard_fast(ls.sz[i], ls.st[i], mu);
            }
            MMPC_PAIR_SETUP
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                MMPC_PAIR(p)
                if (pok) {
                    const double lo = ls.b_lo[p], hi = ls.b_hi[p], vn = XU[idx];
                    if (lo > -1e299) ls.lo_z[p] = mmpc_z_safeguard_fast(ls.lo_z[p], mmpc_box_t(vn - lo), mu);
                    if (hi < 1e299) ls.hi_z[p] = mmpc_z_safeguard_fast(ls.hi_z[p], mmpc_box_t(hi - vn), mu);
                }
            }
            LANES_END
        }
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
            const double sn = TRG[k * F::TRGS], cs = TRG[k * F::TRGS + 1], sks = S[slack_idx(k)];
            double dr[3], dz[3], swr = 0.0;
#pragma unroll
            for (int a = 0; a < 3; a++) { dr[a] = TRG[k * F::TRGS + 2 + a]; dz[a] = TRG[k * F::TRGS + 5 + a]; }
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
    bool leave = false;
    for (;;) {
        bool soc_enter = false;
#pragma unroll 1
        for (;;) {
#define MMPC_ITER_SOC 0
#define MMPC_ITER_LEAVE { leave = true; break; }
#include "mmpc_fast_iter.inc"
#undef MMPC_ITER_SOC
        }
        if (leave || !soc_enter) break;
        // ---- a second-order correction starts: the uncorrected direction and the residuals of the rejected trial go to io.soc,
        //      the point moves back to x_k (state 1)
        soc_dir(true);
        soc_rows(false, false);
        apply_step(-alpha, false);
        alpha0 = alpha; a_prev = alpha; soc_p = 0; soc_st = 1; in_ls = 0;
#pragma unroll 1
        for (;;) {
            if (soc_st == 0) break;   // the correction is over (accepted, or the line search is back on the uncorrected direction)
#define MMPC_ITER_SOC 1
#include "mmpc_fast_iter.inc"
#undef MMPC_ITER_SOC
#undef MMPC_ITER_LEAVE
        }
        if (leave) break;
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
