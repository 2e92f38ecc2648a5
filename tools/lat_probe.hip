// lat_probe.hip - issue / latency figures of the gfx950 instructions the solver's serial chains are made of, measured inside
// one wavefront with s_memtime (cycles of the shader clock).  Build: hipcc --offload-arch=gfx950 -O2 -o lat_probe lat_probe.hip
// Every test is an asm block of REP copies of a short sequence between two s_memtime reads; the empty block is subtracted.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define T0 unsigned long long t0_, t1_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory");
#define T1(slot) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); if (threadIdx.x == 0) out[slot] = (double)(t1_ - t0_);

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)

__global__ void probe(double *out, double seed) {
    __shared__ double lds[1024];
    const int lane = threadIdx.x;
    double a = seed + lane, b = 1.0000001, c = 0.5;
    int ia = lane, ib = 0;
    lds[lane] = a; lds[lane + 64] = b;
    __syncthreads();
    { T0 T1(0) }   // empty
    // 1: 64 independent v_readlane_b32 into two alternating SGPRs (throughput)
    { T0 asm volatile(R16("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 5\n v_readlane_b32 s22, %0, 7\n v_readlane_b32 s23, %0, 9\n") :: "v"(ia) : "s20", "s21", "s22", "s23"); T1(1) }
    // 2: 64 x (readlane -> VALU that reads the SGPR -> readlane of that result): dependent round trip
    { T0 asm volatile(R64("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0\n") : "+v"(ia) :: "s20"); T1(2) }
    // 3: 64 dependent v_fma_f64
    { T0 asm volatile(R64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c)); T1(3) }
    // 4: 64 independent v_fma_f64 (4 accumulators)
    { double a1 = a, a2 = a + 1, a3 = a + 2, a4 = a + 3;
      T0 asm volatile(R16("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                      : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c)); T1(4) a += a1 + a2 + a3 + a4; }
    // 5: 64 dependent v_rcp_f64
    { T0 asm volatile(R64("v_rcp_f64 %0, %0\n") : "+v"(a)); T1(5) }
    // 6: 64 independent v_rcp_f64 (4 values)
    { double a1 = a, a2 = a + 1, a3 = a + 2, a4 = a + 3;
      T0 asm volatile(R16("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n") : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4)); T1(6) a += a1 + a2 + a3 + a4; }
    // 7: 64 x v_cndmask_b32 (independent pairs)
    { int x1 = ia, x2 = ia + 1;
      T0 asm volatile(R16("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n") : "+v"(x1), "+v"(x2) : "v"(ib) : "vcc"); T1(7) ia += x1 + x2; }
    // 8: 16 dependent MFMA f64 16x16x4 (accumulate chain)
    { typedef double v4d __attribute__((ext_vector_type(4)));
      v4d e = {a, a, a, a};
      T0 asm volatile(R16("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n") : "+v"(e) : "v"(b), "v"(c)); T1(8) a += e[0] + e[1] + e[2] + e[3]; }
    // 9: 16 x (MFMA -> VALU reads the result -> feeds the next MFMA's B operand)   (builtins: the compiler inserts the hazard nops)
    { typedef double v4d __attribute__((ext_vector_type(4)));
      v4d e = {a, a, a, a}; double f = c;
      T0
#pragma unroll
      for (int i = 0; i < 16; i++) { e = __builtin_amdgcn_mfma_f64_16x16x4f64(b, f, e, 0, 0, 0); f = e[0] * b; }
      T1(9) a += e[0] + f; }
    // 10: 16 x LDS write -> read of the neighbour's word -> dependent
    { double v = a; int addr = lane * 8, addr2 = ((lane + 1) & 63) * 8;
      T0 asm volatile(R16("ds_write_b64 %1, %0\n ds_read_b64 %0, %2\n s_waitcnt lgkmcnt(0)\n") : "+v"(v) : "v"(addr), "v"(addr2) : "memory"); T1(10) a += v; }
    // 11: 16 dependent ds_read_b64 (address from the previous read: pointer chase, broadcast address)
    { int p = 0; ((int *)lds)[0] = 0; __syncthreads();
      T0 asm volatile(R16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(p) :: "memory"); T1(11) ia += p; }
    // 12: 64 independent ds_read_b64 of one (uniform) address, one wait at the end
    { int p = 64; double v1, v2, v3, v4;
      T0 asm volatile(R16("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n") "s_waitcnt lgkmcnt(0)\n"
                      : "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4) : "v"(p) : "memory"); T1(12) a += v1 + v2 + v3 + v4; }
    // 13: 16 dependent v_permlane16_swap_b32
    { int x1 = ia, x2 = ia + 7;
      T0 asm volatile(R16("v_permlane16_swap_b32 %0, %1\n s_nop 1\n") : "+v"(x1), "+v"(x2)); T1(13) ia += x1 + x2; }
    // 14: 64 dependent v_add_f64
    { T0 asm volatile(R64("v_add_f64 %0, %0, %1\n") : "+v"(a) : "v"(b)); T1(14) }
    // 15: 64 dependent v_mul_f64 alternating with independent v_fma (does in-order issue overlap a dependent chain with independent work?)
    { double a1 = a + 1;
      T0 asm volatile(R64("v_mul_f64 %0, %0, %2\n v_fma_f64 %1, %3, %2, %3\n") : "+v"(a), "+v"(a1) : "v"(b), "v"(c)); T1(15) a += a1; }
    // 16: readlane pairs (one double) x 32 followed by ONE use each at the end (batched broadcasts)
    { double acc = 0;
      T0 asm volatile(R16("v_readlane_b32 s20, %1, 3\n v_readlane_b32 s21, %2, 3\n v_readlane_b32 s22, %1, 4\n v_readlane_b32 s23, %2, 4\n v_add_f64 %0, %0, s[20:21]\n v_add_f64 %0, %0, s[22:23]\n")
                      : "+v"(acc) : "v"(ia), "v"(ib) : "s20", "s21", "s22", "s23"); T1(16) a += acc; }
    // 17: 16 x s_barrier (all waves of the block)
    { T0 asm volatile(R16("s_barrier\n") ::: "memory"); T1(17) }
    // 18: 64 x ds_bpermute_b32 dependent
    { int x1 = ia, ad = ((lane + 1) & 63) * 4;
      T0 asm volatile(R16("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(x1) : "v"(ad) : "memory"); T1(18) ia += x1; }
    if (a == 123.456 || ia == -77) out[63] = a + ia;
}

int main() {
    double *d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    for (int nthreads : {64, 128}) {
        probe<<<1, nthreads>>>(d, 1.5); hipDeviceSynchronize();
        probe<<<1, nthreads>>>(d, 1.5); hipDeviceSynchronize();
        double h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const char *nm[] = {"empty", "v_readlane_b32 independent", "readlane -> VALU use -> readlane (pair)", "v_fma_f64 dependent", "v_fma_f64 independent",
                            "v_rcp_f64 dependent", "v_rcp_f64 independent", "v_cndmask_b32", "MFMA f64 16x16x4 dependent accumulate", "MFMA -> VALU read -> B operand of next",
                            "LDS write -> neighbour read", "ds_read_b32 dependent (pointer chase)", "ds_read_b64 independent, one address", "v_permlane16_swap dependent",
                            "v_add_f64 dependent", "dependent v_mul_f64 + independent v_fma_f64 (pair)", "2 readlane pairs + 2 uses (per 6 instr)", "s_barrier", "ds_bpermute_b32 dependent"};
        const int cnt[] = {1, 64, 64, 64, 64, 64, 64, 64, 16, 16, 16, 16, 64, 16, 64, 64, 16, 16, 16};
        printf("block of %d threads:\n", nthreads);
        for (int i = 1; i <= 18; i++) printf("  %-58s %7.1f cycles each (%d in %5.0f cycles)\n", nm[i], (h[i] - h[0]) / cnt[i], cnt[i], h[i] - h[0]);
    }
    return 0;
}
