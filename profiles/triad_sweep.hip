// Streaming ceiling of the device: triad a = b + s*c and a plain copy, several launch shapes.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U>
__global__ void __launch_bounds__(256) triad(long long n2, double2 *__restrict__ a, const double2 *__restrict__ b, const double2 *__restrict__ c, double s)
{
  const long long stride = (long long)gridDim.x * blockDim.x * U;
  for (long long i0 = (long long)blockIdx.x * blockDim.x * U + threadIdx.x; i0 < n2; i0 += stride) {
    double2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; u++) { long long i = i0 + u * 256; if (i < n2) { x[u] = b[i]; y[u] = c[i]; } }
#pragma unroll
    for (int u = 0; u < U; u++) { long long i = i0 + u * 256; if (i < n2) a[i] = make_double2(x[u].x + s * y[u].x, x[u].y + s * y[u].y); }
  }
}
template <int U>
__global__ void __launch_bounds__(256) copyk(long long n2, double2 *__restrict__ a, const double2 *__restrict__ b)
{
  const long long stride = (long long)gridDim.x * blockDim.x * U;
  for (long long i0 = (long long)blockIdx.x * blockDim.x * U + threadIdx.x; i0 < n2; i0 += stride) {
    double2 x[U];
#pragma unroll
    for (int u = 0; u < U; u++) { long long i = i0 + u * 256; if (i < n2) x[u] = b[i]; }
#pragma unroll
    for (int u = 0; u < U; u++) { long long i = i0 + u * 256; if (i < n2) a[i] = x[u]; }
  }
}
int main()
{
  const long long n = 1LL << 27, n2 = n / 2;
  double *a, *b, *c;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8);
  hipMemset(b, 0, n * 8); hipMemset(c, 0, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {1024, 2048, 4096, 8192, 16384, 65536};
  for (int g : grids) {
#define RUN(NAME, BYTES, ...) { __VA_ARGS__; hipEventRecord(e0); for (int r = 0; r < 20; r++) { __VA_ARGS__; } hipEventRecord(e1); hipEventSynchronize(e1); \
      float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-10s grid %6d: %7.0f GB/s\n", NAME, g, BYTES * (double)n * 20 / (ms * 1e-3) / 1e9); }
    RUN("triad U1", 24., hipLaunchKernelGGL(triad<1>, dim3(g), dim3(256), 0, 0, n2, (double2 *)a, (double2 *)b, (double2 *)c, 0.5))
    RUN("triad U4", 24., hipLaunchKernelGGL(triad<4>, dim3(g), dim3(256), 0, 0, n2, (double2 *)a, (double2 *)b, (double2 *)c, 0.5))
    RUN("copy U4", 16., hipLaunchKernelGGL(copyk<4>, dim3(g), dim3(256), 0, 0, n2, (double2 *)a, (double2 *)b))
  }
  return 0;
}
