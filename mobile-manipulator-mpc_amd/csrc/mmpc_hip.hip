// mmpc_hip.hip - gfx950 kernels + the C ABI of include/mmpc.h (libmmpc.so).
// One 64-lane workgroup (= one wavefront) per problem instance; the solver core is
// mmpc_core.h.  No CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "../../include/mmpc.h"
#include "mmpc_core.h"
#include "mmpc_fast.h"
#include "mmpc_ik.h"

template <int KIND, int NC = 0, int MC = -1, int OPSC = -1, int LC = -1>
__global__ __launch_bounds__(MMPC_WAVE) void mmpc_solve_kernel(
    const MmpcParams *__restrict__ Pp, int B, const double *__restrict__ x_init, const double *__restrict__ traj_ref,
    const double *__restrict__ u_ref, const double *__restrict__ u_last, const double *__restrict__ x_guess,
    const double *__restrict__ obs, double *__restrict__ X, double *__restrict__ U, double *__restrict__ s,
    int *__restrict__ status, int *__restrict__ iters, double *__restrict__ cost, double *__restrict__ err,
    const int *__restrict__ order, double *__restrict__ soc, int soc_stride) {
    extern __shared__ double lds[];
    typedef MmpcDims<KIND> D;
    if ((int)blockIdx.x >= B) return;
    // longest-first schedule hint: workgroup i solves instance order[i] (a permutation; results do not depend on it)
    const int b = order ? order[blockIdx.x] : (int)blockIdx.x;
    const MmpcParams &P = *Pp;
    const int N = NC ? NC : P.N, M = MC >= 0 ? MC : P.M;
    const size_t so = (size_t)((OPSC >= 0 ? OPSC : P.obs_per_stage) ? N + 1 : 1) * M * 3;
    MmpcIO io;
    io.x_init = x_init + (size_t)b * D::NX;
    io.traj_ref = traj_ref + (size_t)b * (N + 1) * D::NREF;
    io.u_ref = u_ref + (size_t)b * N * D::NU;
    io.u_last = u_last + (size_t)b * N * D::NU;
    io.x_guess = x_guess ? x_guess + (size_t)b * (N + 1) * D::NX : nullptr;
    io.u_guess = P.u_guess ? P.u_guess + (size_t)b * N * D::NU : nullptr;
    io.obs = obs + (size_t)b * so;
    io.X = X + (size_t)b * (N + 1) * D::NX;
    io.U = U + (size_t)b * N * D::NU;
    io.s = s + (size_t)b * (N + 1);
    io.status = status + b;
    io.iters = iters + b;
    io.cost = cost + b;
    io.err = err + b;
    io.state = nullptr; io.budget = 0; io.resume = 0; io.gscr = nullptr;   // (iteration budgets are a feature of the specialised kernels)
    io.soc = soc ? soc + (size_t)b * soc_stride : nullptr;
    mmpc_solve_one<KIND, NC, MC, OPSC, LC>(P, io, lds);
}

// WPE = waves per SIMD the register allocation is sized for (2: <= 256 registers, only pays when LDS allows >4 problems/CU)
// The generic kernel for a shape that is a constant of the instantiation (the demo's shapes, MMPC_STATIC_LIST): static LDS block
// and compile-time layout, as for the specialised kernels below; same code, same results as mmpc_solve_kernel<KIND>.
template <int KIND, int NC, int MC, int OPSC, int LC, int AWC>
__global__ __launch_bounds__(MMPC_WAVE) void mmpc_solve_kernel_static(
    const MmpcParams *__restrict__ Pp, int B, const double *__restrict__ x_init, const double *__restrict__ traj_ref,
    const double *__restrict__ u_ref, const double *__restrict__ u_last, const double *__restrict__ x_guess,
    const double *__restrict__ obs, double *__restrict__ X, double *__restrict__ U, double *__restrict__ s,
    int *__restrict__ status, int *__restrict__ iters, double *__restrict__ cost, double *__restrict__ err,
    const int *__restrict__ order, double *__restrict__ soc, int soc_stride) {
    typedef MmpcDims<KIND> D;
    constexpr int NHS = (KIND == 0 && LC > 0) ? 6 : 0, NQ = (KIND == 0 && AWC && LC >= 2) ? 6 * (LC - 1) : 0;
    __shared__ double lds[mmpc_layout<KIND>(NC, MC, OPSC, NHS, NQ).total];
    if ((int)blockIdx.x >= B) return;
    const int b = order ? order[blockIdx.x] : (int)blockIdx.x;
    const MmpcParams &P = *Pp;
    constexpr int N = NC, M = MC;
    constexpr size_t so = (size_t)(OPSC ? N + 1 : 1) * M * 3;
    MmpcIO io;
    io.x_init = x_init + (size_t)b * D::NX;
    io.traj_ref = traj_ref + (size_t)b * (N + 1) * D::NREF;
    io.u_ref = u_ref + (size_t)b * N * D::NU;
    io.u_last = u_last + (size_t)b * N * D::NU;
    io.x_guess = x_guess ? x_guess + (size_t)b * (N + 1) * D::NX : nullptr;
    io.u_guess = P.u_guess ? P.u_guess + (size_t)b * N * D::NU : nullptr;
    io.obs = obs + (size_t)b * so;
    io.X = X + (size_t)b * (N + 1) * D::NX;
    io.U = U + (size_t)b * N * D::NU;
    io.s = s + (size_t)b * (N + 1);
    io.status = status + b;
    io.iters = iters + b;
    io.cost = cost + b;
    io.err = err + b;
    io.state = nullptr; io.budget = 0; io.resume = 0; io.gscr = nullptr;
    io.soc = soc ? soc + (size_t)b * soc_stride : nullptr;
    mmpc_solve_one<KIND, NC, MC, OPSC, LC, AWC>(P, io, lds);
}
// (kind, N, M, obs_per_stage, L, as_written): demo_wholebody_qref.py scenario 2 (two planes) as written and with the intended
// rows, scenario 1 (three planes) as written, the plane-free shape (the terminal-xy 'approach' phase of scenario 0), and the
// shape of BASELINE configs C3 / C4 for the cases the specialised kernel refuses (dense weights, terminal equality): 38 -> 32 ms
// per 8192 against the run-time-sized kernel (no scalar spills: every LDS offset an immediate)
#ifndef MMPC_STATIC_LIST   // (experiments build shorter lists: tools/build_variant.sh)
#define MMPC_STATIC_LIST(X) X(0, 20, 3, 0, 2, 1) X(0, 20, 3, 0, 2, 0) X(0, 20, 3, 0, 3, 1) X(0, 20, 3, 0, 0, 0) X(0, 20, 5, 0, 0, 0)
#endif

// OPS = obstacle table per stage (config obs_per_stage): part of the LDS layout.  The LDS block is STATIC (its size is a
// constant of the instantiation): with `extern __shared__` the base of the dynamic block is resolved after instruction
// selection and every lane-derived LDS address carries an add of that constant 0 (14 of the ~200 instructions of a Riccati stage).
template <int KIND, int N, int MC, int WPE, bool CONT, int OPS>
__global__ __launch_bounds__(MMPC_WAVE, WPE) void mmpc_fast_kernel(
    const MmpcParams *__restrict__ Pp, int B, const double *__restrict__ x_init, const double *__restrict__ traj_ref,
    const double *__restrict__ u_ref, const double *__restrict__ u_last, const double *__restrict__ x_guess,
    const double *__restrict__ obs, double *__restrict__ X, double *__restrict__ U, double *__restrict__ s,
    int *__restrict__ status, int *__restrict__ iters, double *__restrict__ cost, double *__restrict__ err,
    const int *__restrict__ order, int budget, double *__restrict__ state, int state_stride, const int *__restrict__ resume_count,
    const int *__restrict__ list_count, double *__restrict__ gscr, double *__restrict__ soc, int soc_stride) {
#ifndef MMPC_LDS_PAD
#define MMPC_LDS_PAD 0     // (experiments: extra doubles of LDS per problem, to lower the number of resident problems per CU)
#endif
    __shared__ double lds[mmpc_fast_layout<KIND, N>(MC, OPS).total + MMPC_LDS_PAD];
    typedef MmpcDims<KIND> D;
    // A continuation launch (resume_count != null): `order` is the compacted list of the suspended instances, *resume_count its
    // length, and the grid is SMALL (MMPC_RESUME_GRID workgroups that stride over the list): a handful of instances is left,
    // and a grid of B workgroups that almost all exit at once would still have to be dispatched one by one - in a stream of
    // batches that competes with the next batch's launch.
    // A list launch (list_count != null, mmpc_solve_list_device): `order` is the caller's list of instances, *list_count its length
    // (read on the device: the caller need not know it), the grid is the list's capacity.
    const int limit = resume_count ? *resume_count : (list_count ? *list_count : B);
    for (int w = (int)blockIdx.x; w < limit; w += (int)gridDim.x) {
        // launch order: workgroup i solves instance order[i] (a permutation / a list; results do not depend on it)
        const int b = order ? order[w] : w;
        // (a list launch takes its row indices from the caller's device memory, where nothing can have checked them: an index
        //  outside the batch is skipped - it would address the inputs, the outputs and the handle's own save areas)
        if ((unsigned)b >= (unsigned)B) { if (!CONT || !resume_count) break; continue; }
        const MmpcParams &P = *Pp;
        const int M = MC;
        const size_t so = (size_t)(P.obs_per_stage ? N + 1 : 1) * M * 3;
        MmpcIO io;
        io.x_init = x_init + (size_t)b * D::NX;
        io.traj_ref = traj_ref + (size_t)b * (N + 1) * D::NX;
        io.u_ref = u_ref + (size_t)b * N * D::NU;
        io.u_last = u_last + (size_t)b * N * D::NU;
        io.x_guess = x_guess ? x_guess + (size_t)b * (N + 1) * D::NX : nullptr;
        io.u_guess = P.u_guess ? P.u_guess + (size_t)b * N * D::NU : nullptr;
        io.obs = obs + (size_t)b * so;
        io.X = X + (size_t)b * (N + 1) * D::NX;
        io.U = U + (size_t)b * N * D::NU;
        io.s = s + (size_t)b * (N + 1);
        io.status = status + b;
        io.iters = iters + b;
        io.cost = cost + b;
        io.err = err + b;
        io.state = state ? state + (size_t)b * state_stride : nullptr;
        io.budget = budget;
        io.resume = resume_count ? 1 : 0;
        io.gscr = MmpcGainBlock<KIND, N>::ON ? gscr + (size_t)b * MmpcGainBlock<KIND, N>::total : nullptr;
        io.soc = soc ? soc + (size_t)b * soc_stride : nullptr;
        mmpc_solve_fast<KIND, N, MC, CONT, OPS>(P, io, lds);
        if (!CONT || !resume_count) break;      // (one instance per workgroup except in a continuation launch)
        __builtin_amdgcn_s_barrier();           // the next instance reuses the LDS block
    }
}

// (kind, N, M) triples with a specialised kernel; everything else runs the generic kernel.
// BASELINE configs C3/C4 (0,20,5), C5 (0,30,8), C2 (1,15,3); the reference demo (0,20,3).
// The base-only kernel needs 16 KB of LDS per problem (9 problems/CU): it is built for 2 waves/SIMD (+21 % measured).
#define MMPC_RESUME_GRID 256   // workgroups of a continuation launch (they stride over the list of suspended instances)
#ifndef MMPC_FAST_LIST   // (experiments build other lists: -D'MMPC_FAST_LIST(X)=X(0, 10, 5, 2)', tools/occupancy_probe.sh)
#define MMPC_FAST_LIST(X) X(0, 20, 5, 1) X(0, 30, 8, 1) X(0, 20, 3, 1) X(1, 15, 3, 2)
#endif

// Longest-processing-time-first order for the NEXT launch: instances sorted by descending iteration count of
// this one (counting sort, 256 bins, one workgroup).  Kernel time is bounded by the slowest instance; starting
// the slow ones first removes the tail.  In receding-horizon use consecutive ticks have correlated difficulty.
// (One workgroup of MMPC_LPT_THREADS = 256 threads: while another stream's solve launch fills the chip - groups of robots out of
//  phase, fleet.py:run_groups - a workgroup only gets a CU slot where a solver wave has just ended, and four waves fit the one
//  SIMD that frees; the 1024-thread form of round 2 needed two free SIMDs of one CU at once and waited up to 7 ms for them.)
#define MMPC_LPT_THREADS 256
__global__ __launch_bounds__(MMPC_LPT_THREADS) void mmpc_lpt_order(int B, const int *__restrict__ iters, int *__restrict__ order) {
    __shared__ int hist[256], start[256];
    const int t = (int)threadIdx.x;
    for (int v = t; v < 256; v += MMPC_LPT_THREADS) hist[v] = 0;
    __syncthreads();
    for (int i = t; i < B; i += MMPC_LPT_THREADS) { int v = iters[i]; v = v < 0 ? 0 : (v > 255 ? 255 : v); atomicAdd(&hist[v], 1); }
    __syncthreads();
    if (t == 0) { int acc = 0; for (int v = 255; v >= 0; v--) { start[v] = acc; acc += hist[v]; } }
    __syncthreads();
    for (int i = t; i < B; i += MMPC_LPT_THREADS) { int v = iters[i]; v = v < 0 ? 0 : (v > 255 ? 255 : v); order[atomicAdd(&start[v], 1)] = i; }
}

// list of the instances a budgeted launch left suspended (any order), and their number
__global__ __launch_bounds__(256) void mmpc_collect_suspended(int B, const int *__restrict__ status, int *__restrict__ list,
                                                             int *__restrict__ count) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B && status[b] == MMPC_STATUS_SUSPENDED) list[atomicAdd(count, 1)] = b;
}

// the same over a list of instances (list launches): only the listed ones can have been suspended by the launch before
__global__ __launch_bounds__(256) void mmpc_collect_suspended_list(int B, const int *__restrict__ in_list, const int *__restrict__ in_count,
                                                                  const int *__restrict__ status, int *__restrict__ list, int *__restrict__ count) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w < *in_count) { const int b = in_list[w]; if ((unsigned)b < (unsigned)B && status[b] == MMPC_STATUS_SUSPENDED) list[atomicAdd(count, 1)] = b; }
}

// A-priori difficulty of an instance, from its data alone: how close the reference path comes to (or how deep it cuts
// into) an inflated obstacle disc - key = max over stages and obstacles of (r + base radius) - |ref_k - centre|.  The
// iteration count of the interior-point solve correlates with it (0.26 on the C4 generator; the instances that need 40+
// iterations are almost all among the deep cuts), and starting the workgroups in descending key order recovers about
// three quarters of what the exact longest-first order gains over batch order (list-scheduling the measured counts on
// 1024 slots: 192 iterations of wall time in batch order, 168 by this key, 160 exact).  Written as a pseudo iteration
// count 0..255 so that mmpc_lpt_order sorts it.
// With half-space obstacles (whole-body kind, L > 0) a second term: how far the arm's sample points of the START state reach
// into the planes (the largest half-space row value at x_init; [-0.5 m clear .. 0.4 m inside] -> 0..255), the larger of the two
// is the key.  On the demo's shape (2048 starts around the two planes, list-scheduled on 256 slots: 372 iterations of wall time
// in batch order, 250 exact) it gives 258 for the rows as written and 256 for the intended rows (exact there: 219) - the
// circles of that scenario are far away and their key alone is batch order.
__global__ __launch_bounds__(64) void mmpc_difficulty_key(int B, int N, int M, int nref, int obs_per_stage,
                                                          const double *__restrict__ traj_ref, const double *__restrict__ obs,
                                                          int *__restrict__ key, const MmpcParams *__restrict__ Pp,
                                                          const double *__restrict__ x_init, int planes) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    int key_hs = 0;
    if (planes > 0) {
        const MmpcParams &P = *Pp;
        const double *x0 = x_init + (size_t)b * 9;
        double sn, cs, dr[3], dz[3], worst_hs = -1.0e9;
        mmpc_sincos(x0[2], &sn, &cs);
        mmpc_arm_segments(x0[6], x0[7], x0[8], dr, dz);
        for (int i = 0; i < 6; i++) worst_hs = fmax(worst_hs, mmpc_hs_row(P, i, x0[0], x0[1], cs, sn, dr, dz, nullptr));
        const double q2 = (worst_hs + 0.5) * (255.0 / 0.9);
        key_hs = (int)fmin(fmax(q2, 0.0), 255.0);   // (a NaN start: 0 - the solve reports it)
    }
    const double *tr = traj_ref + (size_t)b * (N + 1) * nref;
    const double *ob = obs + (size_t)b * (obs_per_stage ? N + 1 : 1) * M * 3;
    double worst = -1.0e9;
    for (int k = 0; k <= N; k++) {
        const double x = tr[k * nref], y = tr[k * nref + 1];
        const double *o = ob + (obs_per_stage ? (size_t)k * M * 3 : 0);
        for (int m = 0; m < M; m++) {
            const double dx = x - o[3 * m], dy = y - o[3 * m + 1];
            worst = fmax(worst, (o[3 * m + 2] + MMPC_BASE_R) - sqrt(dx * dx + dy * dy));
        }
    }
    // [-1.5 m clearance .. 1.0 m penetration] -> 0..255
    const double q = (worst + 1.5) * (255.0 / 2.5);
    const int key_c = M > 0 ? (int)fmin(fmax(q, 0.0), 255.0) : 0;
    key[b] = key_c > key_hs ? key_c : key_hs;
}

// out_u0[b][a] = U[b][0][a]
__global__ void mmpc_gather_u0(int B, int NU, int stride, const double *__restrict__ U, double *__restrict__ u0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * NU) u0[i] = U[(size_t)(i / NU) * stride + (i % NU)];
}

// Warm-start bookkeeping of the host-pointer entry point.  The reference assigns u_latest / x_guess only after a
// successful solve (mpc_wholebody_qref.py:329-330 come after the exception of :315): a failed instance keeps its
// previous warm start.  warm[b] = 1 once instance b has a converged solution in its rows.
__global__ void mmpc_keep_converged(int B, int nU, int nX, const int *__restrict__ status, const double *__restrict__ U,
                                    const double *__restrict__ X, double *__restrict__ u_latest, double *__restrict__ x_guess,
                                    int *__restrict__ warm) {
    const int b = blockIdx.x;
    if (b >= B || status[b] != MMPC_STATUS_CONVERGED) return;
    for (int i = threadIdx.x; i < nU; i += blockDim.x) u_latest[(size_t)b * nU + i] = U[(size_t)b * nU + i];
    for (int i = threadIdx.x; i < nX; i += blockDim.x) x_guess[(size_t)b * nX + i] = X[(size_t)b * nX + i];
    if (threadIdx.x == 0) warm[b] = 1;
}
// X guess of the instances that have no converged solution yet = what a null guess means: tile(x_init), clipped to xlim
// for the whole-body kinds (:290-291,:302; mpc_base.py:191-207 does not clip)
__global__ void mmpc_cold_xguess(int B, int NS, int NX, int clip, const MmpcParams *__restrict__ P, const int *__restrict__ warm,
                                 const double *__restrict__ x_init, double *__restrict__ x_guess) {
    const int b = blockIdx.x;
    if (b >= B || warm[b]) return;
    for (int i = threadIdx.x; i < NS * NX; i += blockDim.x) {
        const int j = i % NX;
        double v = x_init[(size_t)b * NX + j];
        if (clip) v = fmax(fmin(v, P->xlim[1][j]), P->xlim[0][j]);
        x_guess[(size_t)b * NS * NX + i] = v;
    }
}

struct mmpc_handle_s {
    mmpc_config cfg;
    MmpcParams hp;          // host copy
    MmpcParams *dp;         // device copy
    int nx, nu, nref, lds_bytes;
    int fast;               // 1: specialised kernel exists for (kind, N, M)
    int fast_lds_bytes;
    int gen_static;    // the shape has a static-LDS instantiation of the generic kernel (MMPC_STATIC_LIST)
    int per_cu, fast_per_cu;   // resident workgroups (= problems) per CU the runtime reports for the two kernels
    int diag;               // weights are diagonal (required by the specialised kernel)
    // one stream at a time: launches of a handle share device state (params, schedule hint, warm start), so work on a
    // new stream is ordered after the handle's previous launch through `ev` (recorded after every launch)
    hipEvent_t ev;
    hipStream_t last_stream;
    int ev_valid;
    int hint_on;            // mmpc_set_schedule_hint: 0 batch order, 1 history when there is one / data-derived otherwise, 2 data-derived
    int *d_key;             // a-priori difficulty keys of the launch being prepared
    // iteration budget / continuation (mmpc_set_iteration_budget, mmpc_resume_batch_device)
    int budget;             // iterations a launch may spend on an instance before it is suspended (0: no budget)
    int state_doubles;      // save area per instance
    double *d_state;        // [max_batch][state_doubles], allocated when a budget is first set
    double *d_gscr;         // [max_batch][gscr_doubles]: gain blocks of the specialised kernels of long horizons (MmpcGainBlock), else null
    int gscr_doubles;
    double *d_soc;          // [max_batch][soc_doubles]: scratch of the second-order correction (mmpc_soc_doubles), every kernel
    int soc_doubles;
    int *d_list, *d_count;  // compacted list of the suspended instances of the last launch
    int resume_B;           // batch size of the budgeted launch whose suspended instances can still be continued (0: nothing to resume)
    int no_lpt_env, force_generic_env;   // MMPC_NO_LPT / MMPC_FORCE_GENERIC, read once at create (diagnostics)
    int *d_warm;            // per instance: 1 once a CONVERGED solve has filled its u_latest / x_guess rows
    // device-side state and staging (capacity max_batch)
    double *d_x_init, *d_traj, *d_uref, *d_obs, *d_ulatest, *d_xguess, *d_X, *d_U, *d_s, *d_cost, *d_err, *d_u0;
    int *d_status, *d_iters;
    int *d_order;           // LPT schedule hint from the previous launch (valid for order_B instances)
    int order_B;
    char err[512];
};

static int fail(mmpc_handle h, int code, const char *fmt, const char *a = "", const char *b = "") {
    if (h) snprintf(h->err, sizeof(h->err), fmt, a, b);
    return code;
}
#define HIPCHK(h, call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) return fail(h, MMPC_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static void default_weights(mmpc_handle h) {
    // mpc_wholebody_qref.py:12-16 / mpc_base.py:11-14
    MmpcParams &p = h->hp;
    memset(p.Q2, 0, sizeof(p.Q2)); memset(p.P2, 0, sizeof(p.P2)); memset(p.RW2, 0, sizeof(p.RW2));
    memset(p.R2, 0, sizeof(p.R2)); memset(p.W2, 0, sizeof(p.W2));
    const int nx = h->nx, nu = h->nu;
    if (h->cfg.kind == MMPC_KIND_WHOLEBODY_POSE) {
        // controllers/mpc_wholebody.py:11-15: Q = 5 I4, P = 50 I4 (4x4, leading dimension 4), R, W as the qref controller
        const double r[5] = {0.1, 0.1, 0, 0, 0}, w[5] = {0, 0, 0.1, 0.1, 0.1};
        for (int i = 0; i < 4; i++) { p.Q2[i * 4 + i] = 2 * 5.0; p.P2[i * 4 + i] = 2 * 50.0; }
        for (int i = 0; i < nu; i++) { p.R2[i * nu + i] = 2 * r[i]; p.W2[i * nu + i] = 2 * w[i]; }
    } else if (h->cfg.kind == MMPC_KIND_WHOLEBODY) {
        const double q[9] = {25, 25, 0, 0, 0, 5, 5, 5, 5}, r[5] = {0.1, 0.1, 0, 0, 0}, w[5] = {0, 0, 0.1, 0.1, 0.1};
        for (int i = 0; i < nx; i++) p.Q2[i * nx + i] = p.P2[i * nx + i] = 2 * q[i];
        for (int i = 0; i < nu; i++) { p.R2[i * nu + i] = 2 * r[i]; p.W2[i * nu + i] = 2 * w[i]; }
    } else {
        const double q[6] = {5, 5, 0, 0, 0, 1};
        for (int i = 0; i < nx; i++) p.Q2[i * nx + i] = p.P2[i * nx + i] = 2 * q[i];
        for (int i = 0; i < nu; i++) p.R2[i * nu + i] = 2.0;
    }
    for (int i = 0; i < nu * nu; i++) p.RW2[i] = p.R2[i] + p.W2[i];
    p.S = 1e5;
}

static void update_diag(mmpc_handle h) {
    const MmpcParams &p = h->hp;
    const int nx = h->nx, nu = h->nu;
    int d = 1;
    for (int i = 0; i < nx; i++) for (int j = 0; j < nx; j++) if (i != j && (p.Q2[i * nx + j] != 0.0 || p.P2[i * nx + j] != 0.0)) d = 0;
    for (int i = 0; i < nu; i++) for (int j = 0; j < nu; j++) if (i != j && (p.R2[i * nu + j] != 0.0 || p.W2[i * nu + j] != 0.0)) d = 0;
    h->diag = d;
}

// (waits for the handle's launches in flight - on whatever stream they were made - before the parameter block changes)
static int upload_params(mmpc_handle h) {
    update_diag(h);
    if (h->ev_valid) HIPCHK(h, hipEventSynchronize(h->ev));
    HIPCHK(h, hipMemcpy(h->dp, &h->hp, sizeof(MmpcParams), hipMemcpyHostToDevice));
    return MMPC_OK;
}

#ifdef MMPC_STAMP
// diagnostic build only: read and clear the per-phase cycle accumulators (not part of include/mmpc.h)
extern "C" int mmpc_debug_read_stamps(unsigned long long *out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mmpc_stamp_acc), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mmpc_stamp_acc), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif
#ifdef MMPC_STAMP_GEN
extern "C" int mmpc_debug_read_gstamps(unsigned long long *out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mmpc_gstamp_acc), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mmpc_gstamp_acc), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif
extern "C" const char *mmpc_version(void) { return "mmpc 0.1 (gfx950)"; }
static thread_local char g_err[512] = "";   // errors of the handle-less entry points (mmpc_ik_*)
extern "C" const char *mmpc_last_error(mmpc_handle h) { return h ? h->err : g_err; }
static bool runs_fast(mmpc_handle h) { return h->fast && h->diag && !h->hp.terminal_xy_eq && !h->force_generic_env; }
extern "C" int mmpc_problems_per_cu(mmpc_handle h) { return h ? (runs_fast(h) ? h->fast_per_cu : h->per_cu) : MMPC_E_ARG; }
extern "C" int mmpc_lds_bytes(mmpc_handle h) { return h ? (runs_fast(h) ? h->fast_lds_bytes : h->lds_bytes) : MMPC_E_ARG; }

extern "C" int mmpc_create(const mmpc_config *cfg, mmpc_handle *out) {
    if (!cfg || !out) return MMPC_E_ARG;
    *out = nullptr;
    if (cfg->kind != MMPC_KIND_WHOLEBODY && cfg->kind != MMPC_KIND_BASE && cfg->kind != MMPC_KIND_WHOLEBODY_POSE) return MMPC_E_ARG;
    if (cfg->N < 1 || cfg->N > 63 || cfg->M < 0 || cfg->M > 16 || cfg->max_batch < 1) return MMPC_E_ARG;
    if (cfg->L < 0 || cfg->L > 8 || (cfg->L > 0 && cfg->kind != MMPC_KIND_WHOLEBODY)) return MMPC_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return MMPC_E_NODEVICE;
    mmpc_handle h = new (std::nothrow) mmpc_handle_s();
    if (!h) return MMPC_E_ARG;
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->nx = cfg->kind != MMPC_KIND_BASE ? 9 : 6;
    h->nu = cfg->kind != MMPC_KIND_BASE ? 5 : 2;
    h->nref = cfg->kind == MMPC_KIND_WHOLEBODY_POSE ? 4 : h->nx;   // reference row: endpoint pose (x,y,z,psi) or the state
    *out = h;  // returned even on failure below so that the caller can read the error text
    HIPCHK(h, hipSetDevice(cfg->device));
    MmpcParams &p = h->hp;
    p.N = cfg->N; p.M = cfg->M; p.obs_per_stage = cfg->obs_per_stage ? 1 : 0;
    p.max_iter = cfg->max_iter > 0 ? cfg->max_iter : 2000;   // the reference passes ipopt.max_iter 2000 (mpc_wholebody_qref.py:280)
    p.use_xguess = 0; p.terminal_xy_eq = 0; p.u_guess = nullptr;
    p.dt = cfg->dt; p.tol = cfg->tol > 0 ? cfg->tol : 1e-8; p.mu_init = cfg->mu_init > 0 ? cfg->mu_init : 1.0;
    for (int r = 0; r < 2; r++) {
        for (int j = 0; j < 5; j++) { p.ulim[r][j] = cfg->ulim[r][j]; p.dulim[r][j] = cfg->dulim[r][j]; }
        for (int j = 0; j < 9; j++) p.xlim[r][j] = cfg->xlim[r][j];
    }
    p.L = cfg->L;
    for (int j = 0; j < 8; j++) for (int a = 0; a < 6; a++) p.hs[j][a] = j < cfg->L ? cfg->halfspace[j][a] : 0.0;
    const int nhs = (cfg->kind == MMPC_KIND_WHOLEBODY && cfg->L > 0) ? 6 : 0;
    p.as_written = (cfg->kind == MMPC_KIND_WHOLEBODY && cfg->L >= 2 && cfg->as_written) ? 1 : 0;
    const int nq8 = p.as_written ? 6 * (cfg->L - 1) : 0;
    default_weights(h);
    const MmpcLayout L = cfg->kind == MMPC_KIND_WHOLEBODY ? mmpc_layout<0>(cfg->N, cfg->M, p.obs_per_stage, nhs, nq8)
                       : cfg->kind == MMPC_KIND_BASE    ? mmpc_layout<1>(cfg->N, cfg->M, p.obs_per_stage, 0)
                                                        : mmpc_layout<2>(cfg->N, cfg->M, p.obs_per_stage, 0);
    h->lds_bytes = L.total * (int)sizeof(double);
    if (h->lds_bytes > 160 * 1024) return fail(h, MMPC_E_ARG, "problem needs %s bytes of LDS%s", "more than 163840");
    if (cfg->kind == MMPC_KIND_WHOLEBODY)
        HIPCHK(h, hipFuncSetAttribute((const void *)mmpc_solve_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes));
    else if (cfg->kind == MMPC_KIND_BASE)
        HIPCHK(h, hipFuncSetAttribute((const void *)mmpc_solve_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes));
    else
        HIPCHK(h, hipFuncSetAttribute((const void *)mmpc_solve_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes));
    h->per_cu = 0;
    if (cfg->kind == MMPC_KIND_WHOLEBODY)
        HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&h->per_cu, mmpc_solve_kernel<0>, MMPC_WAVE, h->lds_bytes));
    else if (cfg->kind == MMPC_KIND_BASE)
        HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&h->per_cu, mmpc_solve_kernel<1>, MMPC_WAVE, h->lds_bytes));
    else
        HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&h->per_cu, mmpc_solve_kernel<2>, MMPC_WAVE, h->lds_bytes));
    h->gen_static = 0;
    if (!getenv("MMPC_NO_STATIC_GENERIC")) {
#define MMPC_X(K, NN, MM, OO, LL, AA)                                                                                \
        if (cfg->kind == K && cfg->N == NN && cfg->M == MM && p.obs_per_stage == OO && cfg->L == LL && p.as_written == AA) {         \
            h->gen_static = 1;                                                                                         \
            HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&h->per_cu, mmpc_solve_kernel_static<K, NN, MM, OO, LL, AA>, MMPC_WAVE, 0)); \
        }
        MMPC_STATIC_LIST(MMPC_X)
#undef MMPC_X
    }
    h->fast = 0; h->fast_lds_bytes = 0; h->fast_per_cu = 0;
    {
#define MMPC_X(K, NN, MM, WW)                                                                                        \
        if (cfg->kind == K && cfg->N == NN && cfg->M == MM && cfg->L == 0) {                                                                       \
            h->fast = 1;                                                                                               \
            h->state_doubles = mmpc_fast_state_doubles<K, NN>(MM);                                                     \
            h->gscr_doubles = MmpcGainBlock<K, NN>::total;                                                             \
            h->fast_lds_bytes = mmpc_fast_layout<K, NN>(MM, p.obs_per_stage).total * (int)sizeof(double);         \
            if (p.obs_per_stage) HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&h->fast_per_cu, mmpc_fast_kernel<K, NN, MM, WW, false, 1>, MMPC_WAVE, 0)); \
            else HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&h->fast_per_cu, mmpc_fast_kernel<K, NN, MM, WW, false, 0>, MMPC_WAVE, 0)); \
        }
        MMPC_FAST_LIST(MMPC_X)
#undef MMPC_X
    }
    const size_t B = (size_t)cfg->max_batch, N = (size_t)cfg->N, nx = (size_t)h->nx, nu = (size_t)h->nu;
    const size_t nobs = (size_t)(p.obs_per_stage ? N + 1 : 1) * (size_t)cfg->M * 3;
    HIPCHK(h, hipMalloc(&h->dp, sizeof(MmpcParams)));
    HIPCHK(h, hipMalloc(&h->d_x_init, B * nx * 8));
    HIPCHK(h, hipMalloc(&h->d_traj, B * (N + 1) * nx * 8));
    HIPCHK(h, hipMalloc(&h->d_uref, B * N * nu * 8));
    HIPCHK(h, hipMalloc(&h->d_obs, (B * nobs + 1) * 8));
    HIPCHK(h, hipMalloc(&h->d_ulatest, B * N * nu * 8));
    HIPCHK(h, hipMalloc(&h->d_xguess, B * (N + 1) * nx * 8));
    HIPCHK(h, hipMalloc(&h->d_X, B * (N + 1) * nx * 8));
    HIPCHK(h, hipMalloc(&h->d_U, B * N * nu * 8));
    HIPCHK(h, hipMalloc(&h->d_s, B * (N + 1) * 8));
    HIPCHK(h, hipMalloc(&h->d_cost, B * 8));
    HIPCHK(h, hipMalloc(&h->d_err, B * 8));
    HIPCHK(h, hipMalloc(&h->d_u0, B * nu * 8));
    HIPCHK(h, hipMalloc(&h->d_status, B * 4));
    HIPCHK(h, hipMalloc(&h->d_iters, B * 4));
    HIPCHK(h, hipMalloc(&h->d_order, B * 4));
    HIPCHK(h, hipMalloc(&h->d_warm, B * 4));
    HIPCHK(h, hipMalloc(&h->d_key, B * 4));
    if (h->fast && h->gscr_doubles) HIPCHK(h, hipMalloc(&h->d_gscr, B * (size_t)h->gscr_doubles * 8));
    h->soc_doubles = mmpc_soc_doubles(cfg->N, h->nx, h->nu, L.NR);
    HIPCHK(h, hipMalloc(&h->d_soc, B * (size_t)h->soc_doubles * 8));
    h->order_B = 0;
    h->hint_on = 1;
    h->no_lpt_env = getenv("MMPC_NO_LPT") != nullptr;
    h->force_generic_env = getenv("MMPC_FORCE_GENERIC") != nullptr;
    HIPCHK(h, hipEventCreateWithFlags(&h->ev, hipEventDisableTiming));
    h->ev_valid = 0; h->last_stream = nullptr;
    HIPCHK(h, hipMemset(h->d_ulatest, 0, B * N * nu * 8));
    HIPCHK(h, hipMemset(h->d_xguess, 0, B * (N + 1) * nx * 8));
    HIPCHK(h, hipMemset(h->d_warm, 0, B * 4));
    int rc = upload_params(h);
    if (rc) return rc;
    h->err[0] = 0;
    return MMPC_OK;
}

extern "C" int mmpc_destroy(mmpc_handle h) {
    if (!h) return MMPC_E_ARG;
    void *ptrs[] = {h->dp, h->d_x_init, h->d_traj, h->d_uref, h->d_obs, h->d_ulatest, h->d_xguess, h->d_X, h->d_U,
                    h->d_s, h->d_cost, h->d_err, h->d_u0, h->d_status, h->d_iters, h->d_order, h->d_warm, h->d_key, h->d_state,
                    h->d_list, h->d_count, h->d_gscr, h->d_soc};
    (void)hipSetDevice(h->cfg.device);
    if (h->ev_valid) (void)hipEventSynchronize(h->ev);
    if (h->ev) (void)hipEventDestroy(h->ev);
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete h;
    return MMPC_OK;
}

extern "C" int mmpc_set_weights(mmpc_handle h, const double *Q, const double *R, const double *P, double S, const double *W) {
    if (!h) return MMPC_E_ARG;
    const int nx = h->nx, nu = h->nu;
    MmpcParams &p = h->hp;
    const int nq = h->nref;   // Q, P weigh the reference error: nx x nx, or 4 x 4 for the endpoint-pose reference
    if (Q) for (int i = 0; i < nq; i++) for (int j = 0; j < nq; j++) p.Q2[i * nq + j] = Q[i * nq + j] + Q[j * nq + i];
    if (P) for (int i = 0; i < nq; i++) for (int j = 0; j < nq; j++) p.P2[i * nq + j] = P[i * nq + j] + P[j * nq + i];
    if (R) for (int i = 0; i < nu; i++) for (int j = 0; j < nu; j++) p.R2[i * nu + j] = R[i * nu + j] + R[j * nu + i];
    if (W) for (int i = 0; i < nu; i++) for (int j = 0; j < nu; j++) p.W2[i * nu + j] = W[i * nu + j] + W[j * nu + i];
    if (S >= 0) p.S = S;
    for (int i = 0; i < nu * nu; i++) p.RW2[i] = p.R2[i] + p.W2[i];
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return upload_params(h);
}

extern "C" int mmpc_set_terminal_xy_equality(mmpc_handle h, int on) {
    if (!h) return MMPC_E_ARG;
    if (on && h->cfg.kind == MMPC_KIND_WHOLEBODY_POSE)
        return fail(h, MMPC_E_UNSUPPORTED, "terminal-xy equality%s%s", " is defined on the joint-space reference only (mpc_wholebody_qref.py)");
    h->hp.terminal_xy_eq = on ? 1 : 0;   // handled by the generic kernel (the specialised kernels do not carry it)
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return upload_params(h);
}

extern "C" int mmpc_reset(mmpc_handle h) {
    if (!h) return MMPC_E_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->cfg.max_batch * h->cfg.N * h->nu * 8;
    if (h->ev_valid) HIPCHK(h, hipEventSynchronize(h->ev));
    HIPCHK(h, hipMemset(h->d_ulatest, 0, n));
    HIPCHK(h, hipMemset(h->d_warm, 0, (size_t)h->cfg.max_batch * 4));
    h->order_B = 0;
    h->resume_B = 0;
    return MMPC_OK;
}

static int launch(mmpc_handle h, int B, const double *x_init, const double *traj, const double *uref, const double *ulast,
                  const double *xguess, const double *obs, double *X, double *U, double *s, int *status, int *iters,
                  double *cost, double *err, hipStream_t st, bool resume = false, const int *ulist = nullptr, const int *ucount = nullptr,
                  int ucap = 0) {
    // (the X guess is a launch argument: null = tile(x_init); nothing in the device parameter block changes per launch)
    // a launch on another stream than the previous one waits for it: both touch the handle's schedule hint
    if (h->ev_valid && st != h->last_stream) HIPCHK(h, hipStreamWaitEvent(st, h->ev, 0));
    const bool use_fast = h->fast && h->diag && !h->hp.terminal_xy_eq && !h->force_generic_env;
    // a continuation belongs to the budgeted launch right before it (same B, same buffers): anything in between - another
    // launch, a reset, a change of the budget, a first continuation - leaves the save areas and the list stale
    if (resume && !(use_fast && h->d_state && h->resume_B == B))
        return fail(h, MMPC_E_UNSUPPORTED, "%s%s", "nothing to resume: the handle's last launch was not a budgeted launch of this batch size on a specialised kernel");
    h->resume_B = 0;
    // launch order of the workgroups (results do not depend on it): the iteration counts of the handle's previous launch of
    // this batch size when there are any and the mode allows them, else the a-priori difficulty key of THIS batch's data
    if (ulist && !use_fast) return fail(h, MMPC_E_UNSUPPORTED, "%s%s", "list launches need a specialised kernel (this (kind, N, M, weights) runs the generic one)");
    const bool lpt = !resume && !ulist && h->hint_on && !h->no_lpt_env && B <= h->cfg.max_batch && B > 256;
    const bool history = lpt && h->hint_on == 1 && h->order_B == B;
    const int key_planes = (h->cfg.kind == MMPC_KIND_WHOLEBODY && h->hp.L > 0) ? h->hp.L : 0;
    if (lpt && !history && (h->cfg.M > 0 || key_planes)) {
        hipLaunchKernelGGL(mmpc_difficulty_key, dim3((B + 63) / 64), dim3(64), 0, st, B, h->cfg.N, h->cfg.M, h->nref,
                           h->hp.obs_per_stage, traj, obs, h->d_key, h->dp, x_init, key_planes);
        hipLaunchKernelGGL(mmpc_lpt_order, dim3(1), dim3(MMPC_LPT_THREADS), 0, st, B, h->d_key, h->d_order);
    }
    const int *order = ulist ? ulist : ((history || (lpt && (h->cfg.M > 0 || key_planes))) ? h->d_order : nullptr);
    const int grid = ulist ? ucap : B;
    if (use_fast) {
#define MMPC_LAUNCH_FAST(K, NN, MM, WW, OPS)                                                                              \
            if (resume || h->budget > 0)                                                                                               \
                hipLaunchKernelGGL((mmpc_fast_kernel<K, NN, MM, WW, true, OPS>), dim3(resume ? (B < MMPC_RESUME_GRID ? B : MMPC_RESUME_GRID) : grid), dim3(MMPC_WAVE), 0, st, h->dp, B, \
                                   x_init, traj, uref, ulast, xguess, obs, X, U, s, status, iters, cost, err, resume ? h->d_list : order, \
                                   resume ? 0 : h->budget, h->d_state, h->state_doubles, resume ? h->d_count : (const int *)nullptr,      \
                                   resume ? (const int *)nullptr : ucount, h->d_gscr, h->d_soc, h->soc_doubles);                                                    \
            else                                                                                                                       \
                hipLaunchKernelGGL((mmpc_fast_kernel<K, NN, MM, WW, false, OPS>), dim3(grid), dim3(MMPC_WAVE), 0, st, h->dp, B,           \
                                   x_init, traj, uref, ulast, xguess, obs, X, U, s, status, iters, cost, err, order, 0,                  \
                                   (double *)nullptr, 0, (const int *)nullptr, ucount, h->d_gscr, h->d_soc, h->soc_doubles);
#define MMPC_X(K, NN, MM, WW)                                                                                          \
        if (h->cfg.kind == K && h->cfg.N == NN && h->cfg.M == MM) {                                                   \
            if (h->cfg.obs_per_stage) { MMPC_LAUNCH_FAST(K, NN, MM, WW, 1) } else { MMPC_LAUNCH_FAST(K, NN, MM, WW, 0) }   \
        }
        MMPC_FAST_LIST(MMPC_X)
#undef MMPC_X
#undef MMPC_LAUNCH_FAST
    } else if (h->gen_static) {
#define MMPC_X(K, NN, MM, OO, LL, AA)                                                                                \
        if (h->cfg.kind == K && h->cfg.N == NN && h->cfg.M == MM && h->hp.obs_per_stage == OO && h->cfg.L == LL && h->hp.as_written == AA) \
            hipLaunchKernelGGL((mmpc_solve_kernel_static<K, NN, MM, OO, LL, AA>), dim3(B), dim3(MMPC_WAVE), 0, st, h->dp, B, x_init, traj, \
                               uref, ulast, xguess, obs, X, U, s, status, iters, cost, err, order, h->d_soc, h->soc_doubles);
        MMPC_STATIC_LIST(MMPC_X)
#undef MMPC_X
    } else if (h->cfg.kind == MMPC_KIND_WHOLEBODY)
        hipLaunchKernelGGL(mmpc_solve_kernel<0>, dim3(B), dim3(MMPC_WAVE), h->lds_bytes, st, h->dp, B, x_init, traj, uref,
                           ulast, xguess, obs, X, U, s, status, iters, cost, err, order, h->d_soc, h->soc_doubles);
    else if (h->cfg.kind == MMPC_KIND_BASE)
        hipLaunchKernelGGL(mmpc_solve_kernel<1>, dim3(B), dim3(MMPC_WAVE), h->lds_bytes, st, h->dp, B, x_init, traj, uref,
                           ulast, xguess, obs, X, U, s, status, iters, cost, err, order, h->d_soc, h->soc_doubles);
    else
        hipLaunchKernelGGL(mmpc_solve_kernel<2>, dim3(B), dim3(MMPC_WAVE), h->lds_bytes, st, h->dp, B, x_init, traj, uref,
                           ulast, xguess, obs, X, U, s, status, iters, cost, err, order, h->d_soc, h->soc_doubles);
    HIPCHK(h, hipGetLastError());
    if (use_fast && h->budget > 0 && !resume) {
        // who is suspended: compacted list for mmpc_resume_batch_device
        HIPCHK(h, hipMemsetAsync(h->d_count, 0, 4, st));
        if (ulist) hipLaunchKernelGGL(mmpc_collect_suspended_list, dim3((ucap + 255) / 256), dim3(256), 0, st, B, ulist, ucount, status, h->d_list, h->d_count);
        else hipLaunchKernelGGL(mmpc_collect_suspended, dim3((B + 255) / 256), dim3(256), 0, st, B, status, h->d_list, h->d_count);
        h->resume_B = B;
    }
    if (lpt && h->hint_on == 1) {
        hipLaunchKernelGGL(mmpc_lpt_order, dim3(1), dim3(MMPC_LPT_THREADS), 0, st, B, iters, h->d_order);
        h->order_B = B;
    }
    HIPCHK(h, hipEventRecord(h->ev, st));
    h->ev_valid = 1; h->last_stream = st;
    return MMPC_OK;
}

extern "C" int mmpc_set_iteration_budget(mmpc_handle h, int budget) {
    if (!h || budget < 0) return fail(h, MMPC_E_ARG, "mmpc_set_iteration_budget: %s%s", "budget must be >= 0");
    if (budget > 0 && !h->fast) return fail(h, MMPC_E_UNSUPPORTED, "mmpc_set_iteration_budget: %s%s", "this (kind, N, M) runs the generic kernel, which has no continuation");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (budget > 0 && !h->d_state) {
        const size_t B = (size_t)h->cfg.max_batch;
        if (h->cfg.max_batch > (1 << 20)) return fail(h, MMPC_E_ARG, "%s%s", "iteration budgets need max_batch <= 2^20");
        HIPCHK(h, hipMalloc(&h->d_state, B * (size_t)h->state_doubles * 8));
        HIPCHK(h, hipMalloc(&h->d_list, B * 4));
        HIPCHK(h, hipMalloc(&h->d_count, 4));
        HIPCHK(h, hipMemset(h->d_count, 0, 4));
    }
    if (h->ev_valid) HIPCHK(h, hipEventSynchronize(h->ev));
    h->budget = budget;
    h->resume_B = 0;
    return MMPC_OK;
}

extern "C" int mmpc_resume_batch_device(mmpc_handle h, int B, const double *d_x_init, const double *d_traj_ref,
                                        const double *d_u_ref, const double *d_u_last, const double *d_x_guess,
                                        const double *d_obs, double *d_X, double *d_U, double *d_s, int *d_status,
                                        int *d_iters, double *d_cost, double *d_err, void *stream) {
    if (h && B == 0) return MMPC_OK;   // an empty batch (e.g. the shard of a rank beyond the batch) is a no-op
    if (!h || B < 1 || B > h->cfg.max_batch || !d_x_init || !d_traj_ref || !d_u_ref || !d_u_last || !d_X || !d_U || !d_s ||
        !d_status || !d_iters || !d_cost || !d_err || (h->cfg.M > 0 && !d_obs))
        return fail(h, MMPC_E_ARG, "mmpc_resume_batch_device: %s%s", "bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return launch(h, B, d_x_init, d_traj_ref, d_u_ref, d_u_last, d_x_guess, d_obs ? d_obs : h->d_obs, d_X, d_U, d_s,
                  d_status, d_iters, d_cost, d_err, (hipStream_t)stream, true);
}

extern "C" int mmpc_suspended_count(mmpc_handle h, int *count) {
    if (!h || !count) return MMPC_E_ARG;
    *count = 0;
    if (!h->d_count) return MMPC_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->ev_valid) HIPCHK(h, hipEventSynchronize(h->ev));
    HIPCHK(h, hipMemcpy(count, h->d_count, 4, hipMemcpyDeviceToHost));
    return MMPC_OK;
}

extern "C" int mmpc_set_schedule_hint(mmpc_handle h, int on) {
    if (!h) return MMPC_E_ARG;
    if (on < 0 || on > 2) return fail(h, MMPC_E_ARG, "mmpc_set_schedule_hint: %s%s", "mode must be 0, 1 or 2");
    h->hint_on = on;
    h->order_B = 0;
    return MMPC_OK;
}

extern "C" int mmpc_set_warm_start(mmpc_handle h, const double *d_u_guess, double mu_init) {
    if (!h || !(mu_init > 0.0)) return fail(h, MMPC_E_ARG, "mmpc_set_warm_start: %s%s", "mu_init must be positive");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    h->hp.u_guess = d_u_guess;
    h->hp.mu_init = mu_init;
    return upload_params(h);
}

extern "C" int mmpc_solve_batch_device(mmpc_handle h, int B, const double *d_x_init, const double *d_traj_ref,
                                       const double *d_u_ref, const double *d_u_last, const double *d_x_guess,
                                       const double *d_obs, double *d_X, double *d_U, double *d_s, int *d_status,
                                       int *d_iters, double *d_cost, double *d_err, void *stream) {
    if (h && B == 0) return MMPC_OK;   // an empty batch (e.g. the shard of a rank beyond the batch) is a no-op
    // B > max_batch: the handle's per-instance buffers (launch order, difficulty keys, save areas of suspended solves, list of
    // the suspended, the opt-in U guess) are sized for max_batch instances
    if (!h || B < 1 || B > h->cfg.max_batch || !d_x_init || !d_traj_ref || !d_u_ref || !d_u_last || !d_X || !d_U || !d_s || !d_status ||
        !d_iters || !d_cost || !d_err || (h->cfg.M > 0 && !d_obs))
        return fail(h, MMPC_E_ARG, "mmpc_solve_batch_device: %s%s", "bad argument (B must be in 1..max_batch)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return launch(h, B, d_x_init, d_traj_ref, d_u_ref, d_u_last, d_x_guess, d_obs ? d_obs : h->d_obs, d_X, d_U, d_s,
                  d_status, d_iters, d_cost, d_err, (hipStream_t)stream);
}

extern "C" int mmpc_solve_list_device(mmpc_handle h, int B, const int *d_list, const int *d_count, int capacity, const double *d_x_init,
                                      const double *d_traj_ref, const double *d_u_ref, const double *d_u_last, const double *d_x_guess,
                                      const double *d_obs, double *d_X, double *d_U, double *d_s, int *d_status, int *d_iters,
                                      double *d_cost, double *d_err, void *stream) {
    if (!h || B < 1 || B > h->cfg.max_batch || !d_list || !d_count || capacity < 1 || capacity > B || !d_x_init || !d_traj_ref || !d_u_ref ||
        !d_u_last || !d_X || !d_U || !d_s || !d_status || !d_iters || !d_cost || !d_err || (h->cfg.M > 0 && !d_obs))
        return fail(h, MMPC_E_ARG, "mmpc_solve_list_device: %s%s", "bad argument (B in 1..max_batch, 1 <= capacity <= B)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return launch(h, B, d_x_init, d_traj_ref, d_u_ref, d_u_last, d_x_guess, d_obs ? d_obs : h->d_obs, d_X, d_U, d_s,
                  d_status, d_iters, d_cost, d_err, (hipStream_t)stream, false, d_list, d_count, capacity);
}

extern "C" int mmpc_solve_batch(mmpc_handle h, int B, const double *x_init, const double *traj_ref, const double *u_ref,
                                const double *obs, double *out_u0, double *out_X, double *out_U, double *out_s,
                                int *out_status, int *out_iters, double *out_cost) {
    if (h && B == 0) return MMPC_OK;   // an empty batch (e.g. the shard of a rank beyond the batch) is a no-op
    if (!h || B < 1 || B > h->cfg.max_batch || !x_init || !traj_ref || !u_ref || !out_u0 || (h->cfg.M > 0 && !obs))
        return fail(h, MMPC_E_ARG, "mmpc_solve_batch: %s%s", "bad argument (B must be in 1..max_batch)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t N = (size_t)h->cfg.N, nx = (size_t)h->nx, nu = (size_t)h->nu, b = (size_t)B;
    const size_t nobs = (size_t)(h->hp.obs_per_stage ? N + 1 : 1) * (size_t)h->cfg.M * 3;
    hipStream_t st = 0;
    HIPCHK(h, hipMemcpyAsync(h->d_x_init, x_init, b * nx * 8, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->d_traj, traj_ref, b * (N + 1) * (size_t)h->nref * 8, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->d_uref, u_ref, b * N * nu * 8, hipMemcpyHostToDevice, st));
    if (nobs) HIPCHK(h, hipMemcpyAsync(h->d_obs, obs, b * nobs * 8, hipMemcpyHostToDevice, st));
    // the base kind and the pose-reference kind warm-start X as well (mpc_base.py:194-201, mpc_wholebody.py:134-139); the
    // joint-reference whole-body controller never does (mpc_wholebody_qref.py:301-302)
    const double *xg = nullptr;
    if (h->cfg.kind != MMPC_KIND_WHOLEBODY) {
        if (h->ev_valid && st != h->last_stream) HIPCHK(h, hipStreamWaitEvent(st, h->ev, 0));
        hipLaunchKernelGGL(mmpc_cold_xguess, dim3(B), dim3(64), 0, st, B, (int)N + 1, (int)nx, h->cfg.kind != MMPC_KIND_BASE ? 1 : 0,
                           h->dp, h->d_warm, h->d_x_init, h->d_xguess);
        xg = h->d_xguess;
    }
    int rc = launch(h, B, h->d_x_init, h->d_traj, h->d_uref, h->d_ulatest, xg, h->d_obs, h->d_X, h->d_U, h->d_s,
                    h->d_status, h->d_iters, h->d_cost, h->d_err, st);
    if (rc) return rc;
    if (h->budget > 0 && h->fast && h->d_state) {   // the host-pointer call returns finished solves: continue the suspended ones at once
        rc = launch(h, B, h->d_x_init, h->d_traj, h->d_uref, h->d_ulatest, xg, h->d_obs, h->d_X, h->d_U, h->d_s,
                    h->d_status, h->d_iters, h->d_cost, h->d_err, st, true);
        if (rc && rc != MMPC_E_UNSUPPORTED) return rc;
    }
    // u_latest <- U*, x_guess <- X*  (:329-330), for the instances that converged
    hipLaunchKernelGGL(mmpc_keep_converged, dim3(B), dim3(64), 0, st, B, (int)(N * nu), (int)((N + 1) * nx), h->d_status, h->d_U,
                       h->d_X, h->d_ulatest, h->d_xguess, h->d_warm);
    hipLaunchKernelGGL(mmpc_gather_u0, dim3((B * (int)nu + 255) / 256), dim3(256), 0, st, B, (int)nu, (int)(N * nu), h->d_U, h->d_u0);
    HIPCHK(h, hipMemcpyAsync(out_u0, h->d_u0, b * nu * 8, hipMemcpyDeviceToHost, st));
    if (out_X) HIPCHK(h, hipMemcpyAsync(out_X, h->d_X, b * (N + 1) * nx * 8, hipMemcpyDeviceToHost, st));
    if (out_U) HIPCHK(h, hipMemcpyAsync(out_U, h->d_U, b * N * nu * 8, hipMemcpyDeviceToHost, st));
    if (out_s) HIPCHK(h, hipMemcpyAsync(out_s, h->d_s, b * (N + 1) * 8, hipMemcpyDeviceToHost, st));
    if (out_status) HIPCHK(h, hipMemcpyAsync(out_status, h->d_status, b * 4, hipMemcpyDeviceToHost, st));
    if (out_iters) HIPCHK(h, hipMemcpyAsync(out_iters, h->d_iters, b * 4, hipMemcpyDeviceToHost, st));
    if (out_cost) HIPCHK(h, hipMemcpyAsync(out_cost, h->d_cost, b * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipEventRecord(h->ev, st));
    HIPCHK(h, hipStreamSynchronize(st));
    return MMPC_OK;
}

extern "C" int mmpc_get_u_latest(mmpc_handle h, int B, double *u_latest) {
    if (!h || !u_latest || B < 1 || B > h->cfg.max_batch) return MMPC_E_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->ev_valid) HIPCHK(h, hipEventSynchronize(h->ev));
    HIPCHK(h, hipMemcpy(u_latest, h->d_ulatest, (size_t)B * h->cfg.N * h->nu * 8, hipMemcpyDeviceToHost));
    return MMPC_OK;
}
extern "C" int mmpc_set_u_latest(mmpc_handle h, int B, const double *u_latest) {
    if (!h || !u_latest || B < 1 || B > h->cfg.max_batch) return MMPC_E_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->ev_valid) HIPCHK(h, hipEventSynchronize(h->ev));
    HIPCHK(h, hipMemcpy(h->d_ulatest, u_latest, (size_t)B * h->cfg.N * h->nu * 8, hipMemcpyHostToDevice));
    return MMPC_OK;
}

// ---- batched inverse kinematics of the arm (manipulator_3DoF.py:79-133): one lane per instance, stateless
__global__ __launch_bounds__(256) void mmpc_ik_kernel(int B, const double *__restrict__ q0, const double *__restrict__ target,
                                                      double *__restrict__ q, int *__restrict__ status, int *__restrict__ iters) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double qi[3] = {q0[3 * b], q0[3 * b + 1], q0[3 * b + 2]};
    double qo[3];
    int it = 0;
    const int st = mmpc_ik_solve(qi, target[2 * b], target[2 * b + 1], qo, &it);
    q[3 * b] = qo[0]; q[3 * b + 1] = qo[1]; q[3 * b + 2] = qo[2];
    if (status) status[b] = st;
    if (iters) iters[b] = it;
}

#define IKCHK(call)                                                                                              \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", #call, hipGetErrorString(e_)); return MMPC_E_HIP; } \
    } while (0)

extern "C" int mmpc_ik_batch_device(int device, int B, const double *d_q0, const double *d_target_xz, double *d_q,
                                    int *d_status, int *d_iters, void *stream) {
    if (B < 0 || !d_q0 || !d_target_xz || !d_q) { snprintf(g_err, sizeof(g_err), "mmpc_ik_batch_device: bad argument"); return MMPC_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        snprintf(g_err, sizeof(g_err), "no HIP device %d (this library has no CPU fallback)", device);
        return MMPC_E_NODEVICE;
    }
    if (B == 0) return MMPC_OK;
    IKCHK(hipSetDevice(device));
    hipLaunchKernelGGL(mmpc_ik_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, d_q0, d_target_xz, d_q, d_status, d_iters);
    IKCHK(hipGetLastError());
    return MMPC_OK;
}

extern "C" int mmpc_ik_batch(int device, int B, const double *q0, const double *target_xz, double *out_q, int *out_status,
                             int *out_iters) {
    if (B < 0 || !q0 || !target_xz || !out_q) { snprintf(g_err, sizeof(g_err), "mmpc_ik_batch: bad argument"); return MMPC_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        snprintf(g_err, sizeof(g_err), "no HIP device %d (this library has no CPU fallback)", device);
        return MMPC_E_NODEVICE;
    }
    if (B == 0) return MMPC_OK;
    IKCHK(hipSetDevice(device));
    double *d_q0 = nullptr, *d_t = nullptr, *d_q = nullptr;
    int *d_st = nullptr, *d_it = nullptr;
    int rc = MMPC_OK;
    hipError_t e = hipSuccess;
    if ((e = hipMalloc(&d_q0, (size_t)B * 24)) == hipSuccess && (e = hipMalloc(&d_t, (size_t)B * 16)) == hipSuccess &&
        (e = hipMalloc(&d_q, (size_t)B * 24)) == hipSuccess && (e = hipMalloc(&d_st, (size_t)B * 4)) == hipSuccess &&
        (e = hipMalloc(&d_it, (size_t)B * 4)) == hipSuccess &&
        (e = hipMemcpy(d_q0, q0, (size_t)B * 24, hipMemcpyHostToDevice)) == hipSuccess &&
        (e = hipMemcpy(d_t, target_xz, (size_t)B * 16, hipMemcpyHostToDevice)) == hipSuccess) {
        rc = mmpc_ik_batch_device(device, B, d_q0, d_t, d_q, d_st, d_it, nullptr);
        if (rc == MMPC_OK && (e = hipDeviceSynchronize()) == hipSuccess && (e = hipMemcpy(out_q, d_q, (size_t)B * 24, hipMemcpyDeviceToHost)) == hipSuccess) {
            if (out_status) e = hipMemcpy(out_status, d_st, (size_t)B * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess && out_iters) e = hipMemcpy(out_iters, d_it, (size_t)B * 4, hipMemcpyDeviceToHost);
        }
    }
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "mmpc_ik_batch: %s", hipGetErrorString(e)); rc = MMPC_E_HIP; }
    (void)hipFree(d_q0); (void)hipFree(d_t); (void)hipFree(d_q); (void)hipFree(d_st); (void)hipFree(d_it);
    return rc;
}
