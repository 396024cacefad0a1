// Developer probe: accuracy of v_rcp_f64 and of 1 / 2 Newton steps on gfx950 (prints max relative error vs 1.0/x).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double a = __builtin_amdgcn_rcp(v);
    double e = __builtin_fma(-v, a, 1.0);
    double b = __builtin_fma(a, e, a);
    e = __builtin_fma(-v, b, 1.0);
    double c = __builtin_fma(b, e, b);
    r0[i] = a; r1[i] = b; r2[i] = c;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = ((s >> 11) + 0.5) / 9007199254740992.0; x[i] = std::exp((u - 0.5) * 200.0); }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; i++) {
        long double t = 1.0L / (long double)x[i];
        m0 = std::fmax(m0, (double)fabsl((a[i] - t) / t)); m1 = std::fmax(m1, (double)fabsl((b[i] - t) / t)); m2 = std::fmax(m2, (double)fabsl((c[i] - t) / t));
    }
    std::printf("v_rcp_f64 max rel err %.3e | +1 Newton %.3e | +2 Newton %.3e (eps = 1.1e-16)\n", m0, m1, m2);
    return 0;
}
