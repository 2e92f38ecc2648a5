// occ_proto.hip - cut-down prototype of the specialised kernel's iteration, to MEASURE what a second resident wave per SIMD is worth
// before any rewrite.  One "iteration" of a "problem" is
//   * a serial chain that only one wave can run: 20 Riccati stages of six dependent v_mfma_f64_16x16x4_f64, three elimination legs
//     (accumulator -> v_readlane -> v_rcp_f64 -> three refinement fma -> multiply -> rank-one MFMA), a P store and the operand loads of
//     the next stage through LDS, filler VALU in the MFMA shadows up to the kernel's 171 instructions per stage; then 20 roll-out
//     stages of 60 instructions with 9 v_readlane pairs (figures: DESIGN section 4, profiles/r03_phase_stamps.txt);
//   * lane-parallel work (evaluation, assembly, row steps, trial move): NPAR blocks of two LDS reads -> 28 fma in four chains -> LDS
//     write, calibrated so that chain : parallel = 52 : 48 of ~88 k cycles per iteration at one wave per SIMD, as measured.
// Three decompositions, same total work per problem:
//   base   : one wave per problem, 39.4 KB LDS per problem -> 4 problems = 4 waves per CU (what ships)
//   routeA : one wave per problem, 19.7 KB LDS per problem -> 8 problems = 8 waves per CU (two per SIMD), the prototype needs few registers
//   routeB : two waves per problem, 39.4 KB LDS -> 4 problems = 8 waves per CU; the second wave takes half of the parallel blocks and
//            waits at an s_barrier while the first runs the chain (14 phase boundaries per iteration)
// Build: hipcc --offload-arch=gfx950 -O2 -o occ_proto occ_proto.hip ;  run: ./occ_proto
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double v4d __attribute__((ext_vector_type(4)));
#define NSTAGE 20
#define NITER 19
#define NPHASE 14

#define R2(x) x x
#define R4(x) x x x x
#define R7(x) x x x x x x x
#define R9(x) R4(x) R4(x) x
#define R11(x) R4(x) R4(x) x x x

// filler: VALU instructions independent of the MFMA chain (index arithmetic, selects, copies in the real kernel)
#define FILL11 asm volatile(R11("v_fma_f64 %0, %0, %2, %3\n") "v_fma_f64 %1, %1, %2, %3\n" : "+v"(f0), "+v"(f1) : "v"(b), "v"(c));

__device__ __forceinline__ void chain(double *lds, int lane, double &acc_out) {
    double b = 1.0000001, c = 0.5, f0 = lane, f1 = lane + 1.0;
    v4d e = {1.0 + lane, 2.0, 3.0, 4.0};
    double op[7];
#pragma unroll 1
    for (int k = 0; k < NSTAGE; k++) {
        // operands of the stage (loaded one stage ahead in the kernel: here at the top, 7 independent reads)
#pragma unroll
        for (int r = 0; r < 7; r++) op[r] = lds[(r * 64 + lane + k) & 1023];
        // R1 + R2: two chains of three dependent accumulates
#pragma unroll
        for (int r = 0; r < 6; r++) { e = __builtin_amdgcn_mfma_f64_16x16x4f64(op[r], b, e, 0, 0, 0); FILL11 }
        // R3: three legs  accumulator -> readlane -> rcp -> refinement -> multiply -> rank-one MFMA
#pragma unroll
        for (int leg = 0; leg < 3; leg++) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(e[2]), 42), hi = __builtin_amdgcn_readlane(__double2hiint(e[2]), 42);
            const double d = __hiloint2double(hi, lo | 1) + 3.0;
            const double r0 = __builtin_amdgcn_rcp(d), er = fma(-d, r0, 1.0), r1 = fma(fma(er, er, er), r0, r0);
            const double w = e[3] * r1;
            lds[1024 + ((leg * 64 + lane + 7 * k) & 1023)] = w;
            e = __builtin_amdgcn_mfma_f64_16x16x4f64(-w, e[3], e, 0, 0, 0);
            FILL11
        }
        // P store
#pragma unroll
        for (int r = 0; r < 4; r++) lds[(r * 64 + lane + 5 * k) & 1023] = e[r] * 1e-30;
        e = e * 1e-3 + op[6];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // forward roll-out: 20 stages, 9 readlane pairs + ~40 VALU each, serial
    double x = e[0];
#pragma unroll 1
    for (int k = 0; k < NSTAGE; k++) {
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(x), 2 * q + 1), hi = __builtin_amdgcn_readlane(__double2hiint(x), 2 * q + 1);
            x = fma(__hiloint2double(hi, lo), 1e-9, x * 0.999);
            asm volatile(R4("v_fma_f64 %0, %0, %1, %2\n") : "+v"(f0) : "v"(b), "v"(c));
        }
        lds[(lane + 9 * k) & 1023] = x;
    }
    acc_out += x + f0 + f1;
}

// one block of lane-parallel work: two LDS reads -> 28 fma in four chains -> LDS write
__device__ __forceinline__ void par_block(double *lds, int lane, int i, double &acc) {
    double a0 = lds[(lane + 3 * i) & 1023], a1 = lds[(lane + 64 + 5 * i) & 1023], a2 = a0 + 1.0, a3 = a1 + 1.0;
    const double b = 0.999999, c = 1e-3;
    asm volatile(R7("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    const double s = (a0 + a1) + (a2 + a3);
    lds[1024 + ((lane + 7 * i) & 1023)] = s * 1e-3;
    acc += s;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int WPP, int LDS_DOUBLES, int WPE>
__global__ __launch_bounds__(64 * WPP, WPE) void proto(double *out, int npar, int niter) {
    __shared__ double lds[LDS_DOUBLES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 64 * WPP) lds[i] = 1.0 + 1e-3 * i;
    __syncthreads();
    double acc = 0.0;
#pragma unroll 1
    for (int it = 0; it < niter; it++) {
        // lane-parallel phases: NPHASE - 1 of them before / after the chain, the blocks split over the waves of the problem
        const int per_phase = npar / (NPHASE - 1), mine = per_phase / WPP;
#pragma unroll 1
        for (int ph = 0; ph < NPHASE - 1; ph++) {
#pragma unroll 1
            for (int i = 0; i < mine; i++) par_block(lds, lane, wave * mine + i + ph, acc);
            if (WPP > 1) __syncthreads();
        }
        if (wave == 0) chain(lds, lane, acc);
        if (WPP > 1) __syncthreads();
    }
    out[(size_t)blockIdx.x * 64 * WPP + tid] = acc;
}

template <int WPP, int LDS_DOUBLES, int WPE>
static double run(const char *name, int B, int npar, double *d_out) {
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, proto<WPP, LDS_DOUBLES, WPE>, 64 * WPP, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        proto<WPP, LDS_DOUBLES, WPE><<<B, 64 * WPP>>>(d_out, npar, NITER);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    // cycles per iteration of a resident problem at 2.4 GHz: time x resident problems / (B x iterations)
    const double cyc = best * 1e-3 * 2.4e9 * (per_cu * 256.0) / ((double)B * NITER);
    printf("%-8s waves/problem %d, LDS %6d B, %d problems (%d waves) per CU: %.3f ms per %d problems x %d iterations (%.0f k cycles per iteration and resident problem)\n",
           name, WPP, LDS_DOUBLES * 8, per_cu, per_cu * WPP, best, B, NITER, cyc * 1e-3);
    return best;
}

int main(int argc, char **argv) {
    const int B = 8192;
    const int npar = argc > 1 ? atoi(argv[1]) : 130;   // parallel blocks per iteration (13 phases x 10)
    double *d_out;
    hipMalloc(&d_out, (size_t)B * 128 * sizeof(double));
    printf("parallel blocks per iteration: %d\n", npar);
    const double t0 = run<1, 4928, 1>("base", B, npar, d_out);
    const double tA = run<1, 2464, 2>("routeA", B, npar, d_out);
    const double tA6 = run<1, 3280, 2>("routeA6", B, npar, d_out);   // 26 KB per problem: six problems per CU (two SIMDs host two waves)
    const double tB = run<2, 4928, 1>("routeB", B, npar, d_out);
    printf("gain over base: routeA (two independent waves per SIMD) %.2fx, six problems per CU %.2fx, routeB (two waves per problem) %.2fx\n", t0 / tA, t0 / tA6, t0 / tB);
    // the chain alone and the parallel part alone at one wave per SIMD, for the 52 : 48 calibration (Riccati pass + roll-out 45.5 k of 87.9 k cycles in profiles/r03_phase_stamps.txt)
    const double tc = run<1, 4928, 1>("chain", B, 0, d_out);
    printf("chain share of the base iteration: %.0f %%\n", 100.0 * tc / t0);
    hipFree(d_out);
    return 0;
}
