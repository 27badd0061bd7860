// Development aid: device-to-host and host-to-device copy rate of 700 MB from pageable, registered and runtime-allocated
// host memory.  hipcc -O2 -o pin_rate pin_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double rate(void *h, void *d, size_t n, bool to_host)
{
  hipMemcpy(to_host ? h : d, to_host ? d : h, n, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 3; i++) hipMemcpy(to_host ? h : d, to_host ? d : h, n, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice);
  hipDeviceSynchronize();
  return 3. * n / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e9;
}
int main()
{
  const size_t n = 703040000;
  void *d; hipMalloc(&d, n);
  void *p = malloc(n); memset(p, 1, n);
  printf("pageable   : D2H %.1f GB/s  H2D %.1f GB/s\n", rate(p, d, n, true), rate(p, d, n, false));
  hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
  printf("hipHostRegister(as is, %p): %s\n", p, hipGetErrorString(e));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    const size_t pg = 4096;
    char *lo = (char *)((size_t)p & ~(pg - 1)), *hi = (char *)(((size_t)p + n + pg - 1) & ~(pg - 1));
    e = hipHostRegister(lo, hi - lo, hipHostRegisterDefault);
    printf("hipHostRegister(whole pages): %s\n", hipGetErrorString(e));
  }
  printf("registered : D2H %.1f GB/s  H2D %.1f GB/s\n", rate(p, d, n, true), rate(p, d, n, false));
  void *q; e = hipHostMalloc(&q, n, hipHostMallocDefault);
  printf("hipHostMalloc: %s\n", hipGetErrorString(e));
  printf("runtime    : D2H %.1f GB/s  H2D %.1f GB/s\n", rate(q, d, n, true), rate(q, d, n, false));
  return 0;
}
