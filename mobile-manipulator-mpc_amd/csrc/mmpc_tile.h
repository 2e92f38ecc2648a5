// mmpc_tile.h - pieces shared by the generic (mmpc_core.h) and the specialised (mmpc_fast.h) kernel: lane-state access in the
// two builds (device / host lane emulation), light reciprocals, the v_mfma_f64_16x16x4_f64 tile macros, cross-lane exchanges.
// Included from mmpc_core.h (after MMPC_DEV / LANES_BEGIN are defined).
#pragma once

#ifdef MMPC_EMU
#define MMPC_LS ls_all[lane]
#define MMPC_WR(i) wr_all[lane][i]
// value a field of the lane state holds in lane j (the emulator runs the lanes of a phase one after another: a field that
// is read across lanes and rewritten in the same phase is double-buffered by the parity of the stage)
#define MMPC_LANE_GET(field, j) (ls_all[j].field)
#define MMPC_LANE_XOR16(field) (ls_all[lane ^ 16].field)
#define MMPC_LANE_LOWER16(field) (ls_all[lane & ~16].field)
#define MMPC_FW_SLOTS 2
#define MMPC_WAVE_ANY(x) (x)     // (a flag declared outside the lane loop has been or-ed / min-ed over the lanes by the loop itself)
#define MMPC_FW_SLOT(k) ((k) & 1)
#define LANES_END_REG }
#else
#define MMPC_LS ls_one
#define MMPC_WR(i) wr_one[i]
#define MMPC_LANE_GET(field, j) mmpc_readlane_f64(ls_one.field, j)
#define MMPC_LANE_XOR16(field) mmpc_xor16_f64(ls_one.field, lane)
#define MMPC_LANE_LOWER16(field) mmpc_lower16_f64(ls_one.field)
#define MMPC_FW_SLOTS 1
#define MMPC_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)   // true in every lane if the flag is set in any
#define MMPC_FW_SLOT(k) 0
#define LANES_END_REG }      // end of a phase whose results travel in registers only: no LDS ordering to enforce
#endif


#ifdef MMPC_EMU
MMPC_DEV double mmpc_rcp(double x) { return 1.0 / x; }
MMPC_DEV double mmpc_rsqrt(double x) { return 1.0 / sqrt(x); }
MMPC_DEV double mmpc_rcp3(double x) { return 1.0 / x; }
MMPC_DEV double mmpc_rcp_piv(double x) { return 1.0 / x; }
MMPC_DEV double mmpc_powf(double x, float e) { return (double)exp2f(e * log2f((float)x)); }
MMPC_DEV void mmpc_sched_fence() {}
#else
// v_rcp_f64 / v_rsq_f64 (seeds good to 2^-24, tools/rcp_probe.hip) + one cubic step: 1.1e-16 / 1.4e-16 worst relative error in three
// / five dependent operations; the IEEE division / sqrt sequences are ~3x longer
// (MMPC_RCP_NEWTON 1: one Newton step instead of the cubic one - a dependent operation less per call, relative error e^2 resp. 3/8 e^2
//  <= 3.6e-15; A/B switch, default off: the slacks' reciprocals enter the multipliers' update)
#ifndef MMPC_RCP_NEWTON
#define MMPC_RCP_NEWTON 0
#endif
MMPC_DEV double mmpc_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x), e = fma(-x, r, 1.0);
#if MMPC_RCP_NEWTON
    return fma(e, r, r);
#else
    return fma(fma(e, e, e), r, r);                    // r (1 + e + e^2), e = 1 - x r
#endif
}
MMPC_DEV double mmpc_rcp3(double x) { return mmpc_rcp(x); }
// reciprocal of a pivot of the elimination legs, where every dependent operation is on the critical path of a Riccati stage: the seed
// and ONE Newton step, r (2 - x r): relative error e^2 <= 2^-48 (3.6e-15) in two dependent fma instead of three (MMPC_PIV_NEWTON 0: the
// cubic step of mmpc_rcp).  The factorisation it feeds is backward stable in the usual sense either way: the pivots' own rounding
// through twenty stages of rank-one updates is of that order.
#ifndef MMPC_PIV_NEWTON
#define MMPC_PIV_NEWTON 1
#endif
MMPC_DEV double mmpc_rcp_piv(double x) {
#if MMPC_PIV_NEWTON
    const double r = __builtin_amdgcn_rcp(x), e = fma(-x, r, 1.0);
    return fma(e, r, r);
#else
    return mmpc_rcp(x);
#endif
}
MMPC_DEV double mmpc_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x), e = fma(-(x * y), y, 1.0);
#if MMPC_RCP_NEWTON
    return fma(y * 0.5, e, y);
#endif
    return fma(y * e, fma(0.375, e, 0.5), y);          // y (1 + e/2 + 3 e^2/8), e = 1 - x y^2
}
// x^e in single precision (only used by the filter's switching rule, a heuristic threshold)
MMPC_DEV double mmpc_powf(double x, float e) { return (double)__builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf((float)x)); }
// keeps the scheduler from interleaving independent unrolled bodies (bounds the live registers)
#ifdef MMPC_NO_SCHED_FENCE
MMPC_DEV void mmpc_sched_fence() {}
#else
MMPC_DEV void mmpc_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif
#endif

// product of two small non-negative integers (LDS offsets): v_mul_u32_u24 / v_mad_u32_u24 instead of the quarter-rate 32-bit multiply
#ifdef MMPC_EMU
#define MMPC_MUL24(a, b) ((a) * (b))
#else
#define MMPC_MUL24(a, b) ((int)__umul24((unsigned)(a), (unsigned)(b)))
#endif

// ---- v_mfma_f64_16x16x4_f64: D = A B + C on one wavefront.  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15] and holds
// rows (l>>4) + 4r of column l&15 of C/D in accumulator register r (tools/mfma_probe.hip checks this map on the device).
// An accumulator's register r is therefore K-block r (rows 4r..4r+3) of a B operand as it stands, and - for a
// symmetric matrix - of an A operand: chained products need no lane movement.
#ifdef MMPC_EMU
struct MmpcAcc {
    double v[4];
    double &operator[](int i) { return v[i]; }
    const double &operator[](int i) const { return v[i]; }
};
#define MMPC_MFMA(ACC, AEXPR, BEXPR) {                                                                                  \
    double a_[MMPC_WAVE], b_[MMPC_WAVE];                                                                               \
    for (int lane = 0; lane < MMPC_WAVE; lane++) { auto &ls = ls_all[lane]; a_[lane] = (AEXPR); b_[lane] = (BEXPR); }   \
    for (int lane = 0; lane < MMPC_WAVE; lane++) {                                                                     \
        auto &ls = ls_all[lane];                                                                                       \
        for (int r_ = 0; r_ < 4; r_++) {                                                                               \
            double s_ = ls.ACC[r_];                                                                                    \
            for (int k_ = 0; k_ < 4; k_++) s_ = fma(a_[16 * k_ + (lane >> 4) + 4 * r_], b_[16 * k_ + (lane & 15)], s_); \
            ls.ACC[r_] = s_;                                                                                           \
        }                                                                                                              \
    } }
#define MMPC_MFMA0(ACC, AEXPR, BEXPR) { for (int lane = 0; lane < MMPC_WAVE; lane++) for (int r_ = 0; r_ < 4; r_++) ls_all[lane].ACC[r_] = 0.0; \
    MMPC_MFMA(ACC, AEXPR, BEXPR) }
#else
typedef double MmpcAcc __attribute__((ext_vector_type(4)));
// (first product of a chain: zero accumulator input, no register initialisation)
#define MMPC_MFMA0(ACC, AEXPR, BEXPR) { auto &ls = ls_one; const MmpcAcc z_ = {0.0, 0.0, 0.0, 0.0};                     \
    ls.ACC = __builtin_amdgcn_mfma_f64_16x16x4f64((AEXPR), (BEXPR), z_, 0, 0, 0); }
#define MMPC_MFMA(ACC, AEXPR, BEXPR) { auto &ls = ls_one; ls.ACC = __builtin_amdgcn_mfma_f64_16x16x4f64((AEXPR), (BEXPR), ls.ACC, 0, 0, 0); }
#endif

#ifndef MMPC_EMU
// broadcast of lane j's value to the whole wave through scalar registers (v_readlane_b32 x 2; j is a constant)
MMPC_DEV double mmpc_readlane_f64(double v, int j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), j), hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
    return __hiloint2double(hi, lo);
}
// value of lane (l ^ 16): the 16-lane rows 0 <-> 1 and 2 <-> 3 trade places (v_permlane16_swap_b32 x 2, gfx950; with both
// operands equal it returns {even rows doubled, odd rows doubled})
MMPC_DEV double mmpc_xor16_f64(double v, int lane) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const bool odd = (lane >> 4) & 1;
    return __hiloint2double(odd ? b[0] : b[1], odd ? a[0] : a[1]);
}
// value of lane J of the caller's own row of 16 lanes, in every lane of that row: ONE v_mov_b64_dpp row_newbcast (the only DPP
// control the double-precision pipe takes) instead of two v_readlane_b32 into a scalar pair and the scalar-operand hazards after them
template <int J>
MMPC_DEV double mmpc_rowbcast_f64(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, true); }
template <int J0, int NJ>
MMPC_DEV void mmpc_rowbcast_all(double x, double *out) {
    if constexpr (J0 < NJ) { out[J0] = mmpc_rowbcast_f64<J0>(x); mmpc_rowbcast_all<J0 + 1, NJ>(x, out); }
}
// value of the EVEN row of 16 lanes of the caller's pair of rows (lane & ~16): what the odd rows get is their lower neighbour's
// word, the even rows keep their own (v_permlane16_swap_b32 x 2 without the select of mmpc_xor16_f64)
MMPC_DEV double mmpc_lower16_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]);
}
// Wave-wide reductions as a butterfly over the steps 1, 2, 4, 8, 16, 32 without the LDS crossbar (ds_bpermute: 57 cycles per
// step, tools/lat_probe.hip): the partner's value comes through DPP inside a row of 16 lanes (quad permutes for 1 and 2;
// for 4 and 8 the half-row / row MIRROR - lane 7-i resp. 15-i holds the same value as lane i^4 resp. i^8 at that point, the
// lanes of a reduced group being bitwise equal) and through v_permlane16_swap / v_permlane32_swap across rows.  Every lane
// ends with the same bits (the combine is commutative), which the host emulation reproduces (mmpc_emu_red).
#endif
#ifndef MMPC_RED_DPP
#define MMPC_RED_DPP 1   // wave reductions through DPP / permlane swaps (steps 1..32) instead of ds_bpermute butterflies (steps 32..1)
#endif
#ifndef MMPC_EMU
#if MMPC_RED_DPP
template <int CTRL>
MMPC_DEV double mmpc_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
MMPC_DEV double mmpc_xor32_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const bool up = (mmpc_lane_id() >> 5) & 1;
    return __hiloint2double(up ? b[0] : b[1], up ? a[0] : a[1]);
}
#define MMPC_WAVE_RED(NAME, OP)                                                                              \
MMPC_DEV double NAME(double v) {                                                                             \
    v = OP(v, mmpc_dpp_f64<0xB1>(v));    /* quad_perm [1,0,3,2]: lane ^ 1 */                                  \
    v = OP(v, mmpc_dpp_f64<0x4E>(v));    /* quad_perm [2,3,0,1]: lane ^ 2 */                                  \
    v = OP(v, mmpc_dpp_f64<0x141>(v));   /* row_half_mirror: the other quad of the 8 */                       \
    v = OP(v, mmpc_dpp_f64<0x140>(v));   /* row_mirror: the other half of the row */                          \
    v = OP(v, mmpc_xor16_f64(v, mmpc_lane_id()));                                                             \
    v = OP(v, mmpc_xor32_f64(v));                                                                             \
    return v; }
// FOUR reductions of one kind at once.  v_permlane32_swap / v_permlane16_swap exchange halves / rows BETWEEN two registers, so a
// pair of partials (a, b) needs no copies and no selects: after the swap one register holds (a's lower half, b's lower half), the other
// (a's upper half, b's upper half), and ONE combine leaves a reduced over the two halves in lanes 0-31 and b in lanes 32-63; the same
// over the rows of 16 leaves a, c, b, d - each reduced over its four rows - in rows 0, 1, 2, 3; one in-row butterfly then reduces
// all four, and v_readlane hands each result to the whole wave through a scalar pair.  29 vector instructions instead of 4 x 30.
// Pairing order (the host emulation mirrors it, mmpc_emu_red4): lane l with l + 32, then with l + 16, then the in-row steps 1, 2,
// half-row mirror, row mirror of the single reductions.
#define MMPC_WAVE_RED4(NAME, OP)                                                                              \
MMPC_DEV void NAME(double a, double b, double c, double d, double *out) {                                      \
    auto swp32 = [](double &x, double &y) {                                                                    \
        const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false); \
        const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false); \
        x = __hiloint2double(hi[0], lo[0]); y = __hiloint2double(hi[1], lo[1]); };                           \
    auto swp16 = [](double &x, double &y) {                                                                    \
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false); \
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false); \
        x = __hiloint2double(hi[0], lo[0]); y = __hiloint2double(hi[1], lo[1]); };                           \
    swp32(a, b); double u = OP(a, b);                                                                          \
    swp32(c, d); double w = OP(c, d);                                                                          \
    swp16(u, w); double v = OP(u, w);                                                                          \
    v = OP(v, mmpc_dpp_f64<0xB1>(v));                                                                          \
    v = OP(v, mmpc_dpp_f64<0x4E>(v));                                                                          \
    v = OP(v, mmpc_dpp_f64<0x141>(v));                                                                         \
    v = OP(v, mmpc_dpp_f64<0x140>(v));                                                                         \
    out[0] = mmpc_readlane_f64(v, 0); out[2] = mmpc_readlane_f64(v, 16);                                      \
    out[1] = mmpc_readlane_f64(v, 32); out[3] = mmpc_readlane_f64(v, 48); }
#endif
#endif

// the same butterfly over an array of MMPC_WAVE per-lane partials in LDS (the generic kernel's reductions): device - every lane
// takes its own word and all end with the same bits; host - the same pairing order.  OP: 0 sum, 1 max, 2 min, 3 the running
// maximum of error measures in which a NaN on either side stays (mmpc_max_err)
#ifdef MMPC_EMU
static inline double mmpc_emu_red_arr(const double *a, int op) {
    double v[MMPC_WAVE], w[MMPC_WAVE];
    for (int l = 0; l < MMPC_WAVE; l++) v[l] = a[l];
    for (int q = 0; q < 6; q++) {
        const int o = MMPC_RED_DPP ? (1 << q) : (32 >> q);
        for (int l = 0; l < MMPC_WAVE; l++) {
            const int p = !MMPC_RED_DPP ? (l ^ o) : (o == 4 ? ((l & ~7) | (7 - (l & 7))) : (o == 8 ? ((l & ~15) | (15 - (l & 15))) : (l ^ o)));
            const double x = v[l], y = v[p];
            w[l] = op == 0 ? x + y : (op == 1 ? (x > y ? x : y) : (op == 2 ? (x < y ? x : y) : ((x > y || x != x) ? x : y)));
        }
        for (int l = 0; l < MMPC_WAVE; l++) v[l] = w[l];
    }
    return v[0];
}
#endif
