// Host lane-emulation build of the HIP solver core (TEST ONLY - never part of the product).
// Compiles mobile-manipulator-mpc_amd/csrc/mmpc_core.h with -DMMPC_EMU so that the phase
// structured kernel runs on the CPU: each phase is a loop over the 64 lanes.  Used by
// tests/test_emu_kernel.py to check the kernel logic (and its memory accesses, under ASAN)
// against the oracle before anything is launched on a GPU.  `reverse` runs the lanes of every
// phase in the opposite order: a result that changes exposes an intra-phase data race.
#define MMPC_EMU 1
#include "../../mobile-manipulator-mpc_amd/csrc/mmpc_fast.h"
#include "../../mobile-manipulator-mpc_amd/csrc/mmpc_ik.h"
#include <stdlib.h>
#include <string.h>

template <int KIND>
static void run(const MmpcParams *P, int B, const double *x_init, const double *traj_ref, const double *u_ref,
                const double *u_last, const double *x_guess, const double *obs, double *X, double *U, double *s,
                int *status, int *iters, double *cost, double *err, int reverse) {
    typedef MmpcDims<KIND> D;
    const int N = P->N, M = P->M;
    MmpcLayout L = mmpc_layout<KIND>(N, M, P->obs_per_stage, (KIND == 0 && P->L > 0) ? 6 : 0,
                                     (KIND == 0 && P->L >= 2 && P->as_written) ? 6 * (P->L - 1) : 0);
    const size_t so = (size_t)(P->obs_per_stage ? N + 1 : 1) * M * 3;
    for (int b = 0; b < B; b++) {
        // exact-size heap slab so that ASAN sees any out-of-slab access
        double *lds = (double *)malloc(sizeof(double) * L.total);
        for (int i = 0; i < L.total; i++) lds[i] = NAN;
        MmpcIO io;
        io.x_init = x_init + (size_t)b * D::NX;
        io.traj_ref = traj_ref + (size_t)b * (N + 1) * D::NREF;
        io.u_ref = u_ref + (size_t)b * N * D::NU;
        io.u_last = u_last + (size_t)b * N * D::NU;
        io.x_guess = x_guess ? x_guess + (size_t)b * (N + 1) * D::NX : nullptr;
        io.u_guess = P->u_guess ? P->u_guess + (size_t)b * N * D::NU : nullptr;
        io.obs = obs + (size_t)b * so;
        io.X = X + (size_t)b * (N + 1) * D::NX;
        io.U = U + (size_t)b * N * D::NU;
        io.s = s + (size_t)b * (N + 1);
        io.status = status + b; io.iters = iters + b; io.cost = cost + b; io.err = err + b;
        io.state = nullptr; io.budget = 0; io.resume = 0; io.gscr = nullptr;
        // scratch of the second-order correction (global memory on the device), exact size as the slab
        const int sdn = mmpc_soc_doubles(N, D::NX, D::NU, L.NR);
        double *soc = (double *)malloc(sizeof(double) * sdn);
        for (int i = 0; i < sdn; i++) soc[i] = NAN;
        io.soc = soc;
        MmpcEmu emu = reverse ? MmpcEmu{63, -1, -1} : MmpcEmu{0, 64, 1};
        mmpc_solve_one<KIND>(*P, io, lds, emu);
        free(soc);
        free(lds);
    }
}

template <int KIND, int N, int MC>
static void run_fast(const MmpcParams *P, int B, const double *x_init, const double *traj_ref, const double *u_ref,
                     const double *u_last, const double *x_guess, const double *obs, double *X, double *U, double *s,
                     int *status, int *iters, double *cost, double *err, int reverse, int budget = 0, double *state = nullptr,
                     int resume = 0) {
    typedef MmpcDims<KIND> D;
    const int M = MC;
    MmpcFastLayout L = mmpc_fast_layout<KIND, N>(M, P->obs_per_stage);
    const size_t so = (size_t)(P->obs_per_stage ? N + 1 : 1) * M * 3;
    for (int b = 0; b < B; b++) {
        double *lds = (double *)malloc(sizeof(double) * L.total);
        for (int i = 0; i < L.total; i++) lds[i] = NAN;
        MmpcIO io;
        io.x_init = x_init + (size_t)b * D::NX;
        io.traj_ref = traj_ref + (size_t)b * (N + 1) * D::NX;
        io.u_ref = u_ref + (size_t)b * N * D::NU;
        io.u_last = u_last + (size_t)b * N * D::NU;
        io.x_guess = x_guess ? x_guess + (size_t)b * (N + 1) * D::NX : nullptr;
        io.u_guess = P->u_guess ? P->u_guess + (size_t)b * N * D::NU : nullptr;
        io.obs = obs + (size_t)b * so;
        io.X = X + (size_t)b * (N + 1) * D::NX;
        io.U = U + (size_t)b * N * D::NU;
        io.s = s + (size_t)b * (N + 1);
        io.status = status + b; io.iters = iters + b; io.cost = cost + b; io.err = err + b;
        const int sd = mmpc_fast_state_doubles<KIND, N>(MC);
        io.state = state ? state + (size_t)b * sd : nullptr; io.budget = budget; io.resume = resume;
        if (resume && status[b] != 3) { free(lds); continue; }   // a continuation launch only runs the suspended instances
        MmpcEmu emu = reverse ? MmpcEmu{63, -1, -1} : MmpcEmu{0, 64, 1};
        // gain block of the long horizons (global memory on the device)
        const int gd = MmpcGainBlock<KIND, N>::total;
        double *gscr = gd ? (double *)malloc(sizeof(double) * gd) : nullptr;
        for (int i = 0; i < gd; i++) gscr[i] = NAN;
        io.gscr = gscr;
        const int sdn = mmpc_soc_doubles(N, D::NX, D::NU, MC + D::NSELF);
        double *soc = (double *)malloc(sizeof(double) * sdn);
        for (int i = 0; i < sdn; i++) soc[i] = NAN;
        io.soc = soc;
        if (budget > 0 || resume) mmpc_solve_fast<KIND, N, MC, true>(*P, io, lds, emu); else mmpc_solve_fast<KIND, N, MC, false>(*P, io, lds, emu);
        free(gscr); free(soc);
        free(lds);
    }
}

// fast path (template on the horizon): returns -1 when this (kind, N, M) has no fast instantiation
extern "C" int mmpc_emu_solve_fast(int kind, const MmpcParams *P, int B, const double *x_init, const double *traj_ref,
                                   const double *u_ref, const double *u_last, const double *x_guess, const double *obs,
                                   double *X, double *U, double *s, int *status, int *iters, double *cost, double *err,
                                   int reverse) {
#define MMPC_FAST_CASE(K, NN, MM) if (kind == K && P->N == NN && P->M == MM) { run_fast<K, NN, MM>(P, B, x_init, traj_ref, u_ref, u_last, x_guess, obs, X, U, s, status, iters, cost, err, reverse); return 0; }
    MMPC_FAST_CASE(0, 20, 5) MMPC_FAST_CASE(0, 20, 3) MMPC_FAST_CASE(0, 30, 8) MMPC_FAST_CASE(0, 20, 0) MMPC_FAST_CASE(1, 15, 3)
#undef MMPC_FAST_CASE
    return -1;
}
// same with an iteration budget: state = [B][mmpc_emu_fast_state_doubles] save area; resume != 0 continues the instances whose
// status[] is 3 (MMPC_STATUS_SUSPENDED) and leaves the others alone
extern "C" int mmpc_emu_solve_fast_budget(int kind, const MmpcParams *P, int B, const double *x_init, const double *traj_ref,
                                          const double *u_ref, const double *u_last, const double *x_guess, const double *obs,
                                          double *X, double *U, double *s, int *status, int *iters, double *cost, double *err,
                                          int budget, double *state, int resume) {
#define MMPC_FAST_CASE(K, NN, MM) if (kind == K && P->N == NN && P->M == MM) { run_fast<K, NN, MM>(P, B, x_init, traj_ref, u_ref, u_last, x_guess, obs, X, U, s, status, iters, cost, err, 0, budget, state, resume); return 0; }
    MMPC_FAST_CASE(0, 20, 5) MMPC_FAST_CASE(1, 15, 3)
#undef MMPC_FAST_CASE
    return -1;
}
extern "C" int mmpc_emu_fast_state_doubles(int kind, int N, int M) {
    if (kind == 0 && N == 20) return mmpc_fast_state_doubles<0, 20>(M);
    if (kind == 1 && N == 15) return mmpc_fast_state_doubles<1, 15>(M);
    return -1;
}
extern "C" int mmpc_emu_fast_lds_doubles(int kind, int N, int M, int obs_per_stage) {
    if (kind == 0 && N == 20) return mmpc_fast_layout<0, 20>(M, obs_per_stage).total;
    if (kind == 0 && N == 30) return mmpc_fast_layout<0, 30>(M, obs_per_stage).total;
    if (kind == 1 && N == 15) return mmpc_fast_layout<1, 15>(M, obs_per_stage).total;
    return -1;
}

extern "C" int mmpc_emu_solve(int kind, const MmpcParams *P, int B, const double *x_init, const double *traj_ref,
                              const double *u_ref, const double *u_last, const double *x_guess, const double *obs,
                              double *X, double *U, double *s, int *status, int *iters, double *cost, double *err,
                              int reverse) {
    if (kind == 0) run<0>(P, B, x_init, traj_ref, u_ref, u_last, x_guess, obs, X, U, s, status, iters, cost, err, reverse);
    else if (kind == 1) run<1>(P, B, x_init, traj_ref, u_ref, u_last, x_guess, obs, X, U, s, status, iters, cost, err, reverse);
    else run<2>(P, B, x_init, traj_ref, u_ref, u_last, x_guess, obs, X, U, s, status, iters, cost, err, reverse);
    return 0;
}
extern "C" int mmpc_emu_lds_doubles(int kind, int N, int M, int obs_per_stage) {
    return kind == 0 ? mmpc_layout<0>(N, M, obs_per_stage).total : kind == 1 ? mmpc_layout<1>(N, M, obs_per_stage).total : mmpc_layout<2>(N, M, obs_per_stage).total;
}
extern "C" int mmpc_emu_params_size() { return (int)sizeof(MmpcParams); }

// arm inverse kinematics (mmpc_ik.h is plain scalar code: the same function the GPU kernel calls per lane)
extern "C" void mmpc_emu_ik(int B, const double *q0, const double *target, double *q, int *status, int *iters) {
    for (int b = 0; b < B; b++) status[b] = mmpc_ik_solve(q0 + 3 * b, target[2 * b], target[2 * b + 1], q + 3 * b, iters + b);
}
