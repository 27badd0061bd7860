// Issue-rate calibration of fp64 VALU instructions on one SIMD (development aid).
//   hipcc --offload-arch=gfx950 -O2 -o ubench_fp64 ubench_fp64.hip && ./ubench_fp64
// Each lane runs N rounds of 8 independent chains of one operation; one workgroup of 64 x W threads
// per CU places W/4 waves on every SIMD.  Prints cycles per wave-instruction as seen by one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void k(double *out, long long *cyc, int n, double a, double b)
{
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
#define STEP(x)                                                          \
    if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(a)); \
    if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b)); \
    if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a)); \
    if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(*(float *)&x) : "v"((float)a));
    STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, int waves_per_simd)
{
  const int n = 4096, nblk = 256, nthr = 256 * waves_per_simd;
  double *out; long long *cyc;
  hipMalloc(&out, sizeof(double) * nblk * nthr);
  hipMalloc(&cyc, sizeof(long long) * nblk);
  hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(nthr), 0, 0, out, cyc, n, 1.0000001, 0.9999999);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(nthr), 0, 0, out, cyc, n, 1.0000001, 0.9999999);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(nblk);
  hipMemcpy(h.data(), cyc, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
  double c = 0; for (auto v : h) c += v; c /= nblk;
  printf("%-10s %d wave(s)/SIMD: %6.2f clock64 ticks per wave-instruction, kernel %.3f ms -> %.2f ns per instr per SIMD slot\n",
         name, waves_per_simd, c / (8.0 * n), ms, ms * 1e6 / (8.0 * n * waves_per_simd));
  hipFree(out); hipFree(cyc);
}

int main()
{
  for (int w = 1; w <= 4; w++) {
    run<0>("v_add_f64", w); run<1>("v_mul_f64", w); run<2>("v_fma_f64", w); run<3>("v_add_f32", w);
  }
  return 0;
}
