// Probe of v_mfma_f64_16x16x4_f64 on gfx950: operand / result lane maps (exact integer data) and the
// latency of a dependent accumulate chain.  Build: hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe(const double *A, const double *B, const double *C, double *D) {
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    // A is 16x4 (row-major), B is 4x16, C/D 16x16
    const double a = A[(l & 15) * 4 + g], b = B[g * 16 + j];
    v4d c;
    for (int r = 0; r < 4; r++) c[r] = C[(g + 4 * r) * 16 + j];
    v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(g + 4 * r) * 16 + j] = d[r];
}

// latency: n dependent MFMAs in one wave, timed from the host
__global__ void chain_acc(int n, double a, double b, double *out) {
    v4d e = {1, 2, 3, 4};
    for (int i = 0; i < n; i++) e = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e, 0, 0, 0);
    out[threadIdx.x] = e[0] + e[1] + e[2] + e[3];
}
// the result feeds the B operand of the next group (register r = k-block r), 4 MFMAs per group
__global__ void chain_bop(int n, double a, double *out) {
    v4d f = {1, 2, 3, 4};
    for (int i = 0; i < n; i++) {
        v4d z = {0, 0, 0, 0};
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f[0], z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f[1], z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f[2], z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f[3], z, 0, 0, 0);
        f = z;
    }
    out[threadIdx.x] = f[0] + f[1] + f[2] + f[3];
}

// one input-elimination step of the Riccati stage: pivot from a lane of the accumulator (v_readlane), reciprocal, scaled row
// as the only non-zero K-slot, rank-one MFMA back onto the accumulator
__global__ void chain_pivot(int n, double *out) {
    v4d e = {1.0 + threadIdx.x, 2, 3, 4};
    const bool own = (threadIdx.x >> 4) == 2;
    for (int i = 0; i < n; i++) {
        const double c = e[2];
        const int lo = __builtin_amdgcn_readlane(__double2loint(c), 42), hi = __builtin_amdgcn_readlane(__double2hiint(c), 42);
        const double d = __hiloint2double(hi, lo) + 1e3;
        const double r = __builtin_amdgcn_rcp(d), q = fma(-d, r, 1.0), id = fma(fma(q, q, q), r, r);
        const double w = c * id;
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(own ? -w : 0.0, own ? c : 0.0, e, 0, 0, 0);
    }
    out[threadIdx.x] = e[0] + e[1] + e[2] + e[3];
}
// the small-tile instruction: v_mfma_f64_4x4x4 (four 4x4x4 blocks), dependent accumulate chain
__global__ void chain_acc4(int n, double a, double b, double *out) {
    double e = threadIdx.x;
    for (int i = 0; i < n; i++) e = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e, 0, 0, 0);
    out[threadIdx.x] = e;
}

// dependent fp64 FMA chain and dependent rcp chain (one wave)
__global__ void chain_fma(int n, double a, double b, double *out) {
    double x = threadIdx.x;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) x = fma(x, a, b);
    }
    out[threadIdx.x] = x;
}
__global__ void chain_rcp(int n, double *out) {
    double x = 1.5 + threadIdx.x;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) x = __builtin_amdgcn_rcp(x);
    }
    out[threadIdx.x] = x;
}
// LDS round trip: write own slot, read the neighbour's, dependent
__global__ void chain_lds(int n, double *out) {
    __shared__ double buf[64];
    double x = threadIdx.x;
    const int nb = (threadIdx.x + 1) & 63;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) { buf[threadIdx.x] = x; __syncthreads(); x = buf[nb] + 1.0; __syncthreads(); }
    }
    out[threadIdx.x] = x;
}

int main() {
    double hA[64], hB[64], hC[256], hD[512], ref[256];
    srand(1);
    for (int i = 0; i < 64; i++) { hA[i] = rand() % 17 - 8; hB[i] = rand() % 13 - 6; }
    for (int i = 0; i < 256; i++) hC[i] = rand() % 11 - 5;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        double s = hC[i * 16 + j];
        for (int k = 0; k < 4; k++) s += hA[i * 4 + k] * hB[k * 16 + j];
        ref[i * 16 + j] = s;
    }
    double *dA, *dB, *dC, *dD; long long *dc, hc[2];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD); hipMalloc(&dc, 16);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dC, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) bad += hD[i] != ref[i];
    printf("layout mismatches: %d of 256\n", bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 1 << 20; float ms;
    chain_acc<<<1, 64>>>(1000, 1e-3, 1e-3, dD); hipDeviceSynchronize();
    hipEventRecord(e0); chain_acc<<<1, 64>>>(n, 1e-3, 1e-3, dD); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("dependent accumulate chain: %.2f ns per MFMA\n", ms * 1e6 / n);
    hipEventRecord(e0); chain_bop<<<1, 64>>>(n / 4, 1e-3, dD); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("result-as-B-operand chain: %.2f ns per MFMA (groups of 4)\n", ms * 1e6 / n);
    hipEventRecord(e0); chain_pivot<<<1, 64>>>(n / 4, dD); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("pivot step (readlane, rcp3, scale, select, rank-one MFMA): %.2f ns per step\n", ms * 1e6 / (n / 4));
    hipEventRecord(e0); chain_acc4<<<1, 64>>>(n, 1e-3, 1e-3, dD); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("dependent 4x4x4 accumulate chain: %.2f ns per MFMA\n", ms * 1e6 / n);
    const int m = 1 << 16;
    chain_fma<<<1, 64>>>(10, 0.5, 1.0, dD); hipDeviceSynchronize();
    hipEventRecord(e0); chain_fma<<<1, 64>>>(m, 0.5, 1.0, dD); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("dependent v_fma_f64 chain: %.2f ns per op\n", ms * 1e6 / (16.0 * m));
    hipEventRecord(e0); chain_rcp<<<1, 64>>>(m, dD); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("dependent v_rcp_f64 chain: %.2f ns per op\n", ms * 1e6 / (16.0 * m));
    hipEventRecord(e0); chain_lds<<<1, 64>>>(m, dD); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("LDS write -> neighbour read -> add round trip: %.2f ns\n", ms * 1e6 / (16.0 * m));
    return bad != 0;
}
