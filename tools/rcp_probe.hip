// accuracy of the v_rcp_f64 / v_rsq_f64 seeds and of the refinement steps the kernels use (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
__global__ void k(const double *x, double *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double r = __builtin_amdgcn_rcp(v), e = fma(-v, r, 1.0);
    o[i] = r;                                     // seed
    o[n + i] = fma(fma(e, e, e), r, r);           // cubic
    double r2 = fma(e, r, r);                     // one Newton
    o[2 * n + i] = r2;
    const double y = __builtin_amdgcn_rsq(v);
    o[3 * n + i] = y;
    const double vy = v * y, e2 = fma(-vy, y, 1.0);
    o[4 * n + i] = fma(y * e2, fma(0.375, e2, 0.5), y);   // cubic
    double y1 = y * fma(-0.5 * v * y, y, 1.5);
    o[5 * n + i] = y1;                                     // one Newton
    o[6 * n + i] = y1 * fma(-0.5 * v * y1, y1, 1.5);        // two Newton
}
int main() {
    const int n = 1 << 20;
    double *hx = (double *)malloc(n * 8), *ho = (double *)malloc(7 * n * 8), *dx, *dout;
    srand(1);
    for (int i = 0; i < n; i++) { double u = rand() / (double)RAND_MAX, s = rand() / (double)RAND_MAX; hx[i] = (0.5 + u) * pow(10.0, -12 + 24 * s); }
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 7 * n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, 7 * n * 8, hipMemcpyDeviceToHost);
    const char *nm[7] = {"rcp seed", "rcp cubic", "rcp 1 Newton", "rsq seed", "rsq cubic", "rsq 1 Newton", "rsq 2 Newton"};
    for (int j = 0; j < 7; j++) {
        double m = 0;
        for (int i = 0; i < n; i++) {
            long double ex = j < 3 ? 1.0L / (long double)hx[i] : 1.0L / sqrtl((long double)hx[i]);
            double re = fabs((double)(((long double)ho[j * n + i] - ex) / ex));
            if (re > m) m = re;
        }
        printf("%-14s max rel err %.3e (2^%.1f)\n", nm[j], m, log2(m));
    }
    return 0;
}
