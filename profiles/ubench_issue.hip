// Issue-rate calibration of fp64 VALU instructions on one SIMD with long straight-line bodies (development aid).
//   hipcc --offload-arch=gfx950 -O2 -o ubench_issue ubench_issue.hip && ./ubench_issue
// Each lane runs N rounds of a body of 256 instructions: CH independent chains (CH = 8: no dependent pair closer
// than 8 instructions, CH = 2: every second instruction depends on the one two before it).  One workgroup of
// 64 x 4 x W threads per CU places W waves on every SIMD.  Reports nanoseconds per wave-instruction per SIMD from the
// kernel time, which is independent of the clock the counters run at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP, int CH>
__global__ void k(double *out, int n, double a, double b)
{
  double x[8];
  for (int c = 0; c < 8; c++) x[c] = threadIdx.x + c;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int r = 0; r < 256 / CH; r++) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
        if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(b), "v"(a));
      }
    }
  }
  double s = 0;
  for (int c = 0; c < 8; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int CH>
void run(const char *name, int waves_per_simd)
{
  const int n = 256, nblk = 256, nthr = 256 * waves_per_simd;
  double *out;
  hipMalloc(&out, sizeof(double) * nblk * nthr);
  hipLaunchKernelGGL((k<OP, CH>), dim3(nblk), dim3(nthr), 0, 0, out, n, 1.0000001, 0.9999999);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, CH>), dim3(nblk), dim3(nthr), 0, 0, out, n, 1.0000001, 0.9999999);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ninstr = 256.0 * n * waves_per_simd;   // wave-instructions per SIMD
  printf("%-10s chains=%d %d wave(s)/SIMD: kernel %.3f ms -> %.3f ns per wave-instruction per SIMD (4 cycles at 2.4 GHz = 1.667 ns)\n",
         name, CH, waves_per_simd, ms, ms * 1e6 / ninstr);
  hipFree(out);
}

int main()
{
  for (int w = 1; w <= 4; w++) {
    run<0, 8>("v_add_f64", w); run<1, 8>("v_mul_f64", w); run<2, 8>("v_fma_f64", w);
    run<0, 2>("v_add_f64", w); run<0, 1>("v_add_f64", w);
  }
  return 0;
}
