// accuracy of v_rsq_f64 / v_rcp_f64 and of one / two Newton steps on them (what the Cholesky pivots in opt.hip and opt_math.h's m_rcp / m_rsqrt rely on)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void kr(const double *in, double *o0, double *o1, double *o2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = in[i], y = __builtin_amdgcn_rcp(d);
    o0[i] = y;
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y); o1[i] = y;
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y); o2[i] = y;
}
__global__ void k(const double *in, double *o0, double *o1, double *o2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = in[i], y = __builtin_amdgcn_rsq(d);
    o0[i] = y;
    y = y * (1.5 - 0.5 * d * y * y); o1[i] = y;
    y = y * (1.5 - 0.5 * d * y * y); o2[i] = y;
}
int main() {
    const int n = 1 << 20;
    double *h = new double[n], *r0 = new double[n], *r1 = new double[n], *r2 = new double[n];
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = std::ldexp(1.0 + (double)(s >> 12) / (double)(1ull << 52), (int)(s % 80) - 40); }
    double *d, *o0, *o1, *o2;
    hipMalloc(&d, n * 8); hipMalloc(&o0, n * 8); hipMalloc(&o1, n * 8); hipMalloc(&o2, n * 8);
    hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o0, o1, o2, n);
    hipMemcpy(r0, o0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1, o1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        const long double ex = 1.0L / sqrtl((long double)h[i]);
        e0 = fmax(e0, (double)fabsl((r0[i] - ex) / ex)); e1 = fmax(e1, (double)fabsl((r1[i] - ex) / ex)); e2 = fmax(e2, (double)fabsl((r2[i] - ex) / ex));
    }
    printf("max relative error: v_rsq_f64 %.3e, + 1 Newton %.3e, + 2 Newton %.3e\n", e0, e1, e2);
    hipLaunchKernelGGL(kr, dim3(n / 256), dim3(256), 0, 0, d, o0, o1, o2, n);
    hipMemcpy(r0, o0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1, o1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, n * 8, hipMemcpyDeviceToHost);
    e0 = e1 = e2 = 0;
    for (int i = 0; i < n; i++) {
        const long double ex = 1.0L / (long double)h[i];
        e0 = fmax(e0, (double)fabsl((r0[i] - ex) / ex)); e1 = fmax(e1, (double)fabsl((r1[i] - ex) / ex)); e2 = fmax(e2, (double)fabsl((r2[i] - ex) / ex));
    }
    printf("max relative error: v_rcp_f64 %.3e, + 1 Newton %.3e, + 2 Newton %.3e\n", e0, e1, e2);
    return 0;
}
