// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS code's access pattern (MI355X_MICROARCH.md, "HBM":
// on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads; other widths are uncalibrated).
// The eval kernels read x / y / log x with coalesced 8-byte-per-lane loads.  This program streams a known number of
// bytes the same way (and, for comparison, 16 B per lane) so that tools/profile_round.sh can divide the counters.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o gpurun_out/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE -d <out> -- gpurun_out/fetch_calib        (and a second pass with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read8(const double *__restrict__ a, size_t n, double *out)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 123.456) out[0] = s;      // never true: keeps the loads
}

__global__ void read16(const double2 *__restrict__ a, size_t n, double *out)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}

__global__ void write8(double *__restrict__ a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}

int main()
{
    const size_t n = (size_t)96 << 20;            // 96 Mi doubles = 768 MiB: three times the Infinity Cache
    double *a = nullptr, *out = nullptr;
    if (hipMalloc(&a, n * sizeof(double)) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(a, 0, n * sizeof(double));
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, a, n, out);
        hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const double2 *>(a), n / 2, out);
        hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, a, n);
    }
    (void)hipDeviceSynchronize();
    printf("bytes per launch: %zu\n", n * sizeof(double));
    return 0;
}
