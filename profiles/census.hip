// Residency census (development aid): how many workgroups of a given shape (threads, registers, LDS) share a CU.
//   hipcc --offload-arch=gfx950 -O2 -w -o census census.hip && ./census
// Every workgroup spins for a fixed time; a grid of g x 256 workgroups takes ceil(g / resident-per-CU) spins.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NT, int WPE, int LDSB>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
spin(long long ticks, int *out)
{
  __shared__ char lds[LDSB];
  // touch a high register so that the allocation is the one of the kernel under study
  if (WPE == 3) asm volatile("v_mov_b32 v166, 0" ::: "v166");
  if (WPE == 2) asm volatile("v_mov_b32 v220, 0" ::: "v220");
  if (WPE == 4) asm volatile("v_mov_b32 v126, 0" ::: "v126");
  lds[threadIdx.x] = 1;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (threadIdx.x == 0) out[blockIdx.x] = lds[0];
}

template <int NT, int WPE, int LDSB>
void run(const char *what)
{
  int *out; hipMalloc(&out, sizeof(int) * 4096);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const long long ticks = 20000;   // 100 MHz wall clock: 200 us
  for (int g = 1; g <= 4; g++) {
    hipLaunchKernelGGL((spin<NT, WPE, LDSB>), dim3(256 * g), dim3(NT), 0, 0, ticks, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((spin<NT, WPE, LDSB>), dim3(256 * g), dim3(NT), 0, 0, ticks, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s grid %4d workgroups: %.3f ms = %.1f spins\n", what, 256 * g, ms, ms / 0.2);
  }
  hipFree(out);
}

int main()
{
  run<256, 2, 74048>("256 threads, 2 waves/SIMD regs, 74 KB LDS");
  run<384, 3, 74048>("384 threads, 3 waves/SIMD regs, 74 KB LDS");
  run<256, 3, 53000>("256 threads, 3 waves/SIMD regs, 53 KB LDS");
  run<512, 4, 74048>("512 threads, 4 waves/SIMD regs, 74 KB LDS");
  run<320, 3, 74048>("320 threads, 3 waves/SIMD regs, 74 KB LDS");
  run<768, 3, 150000>("768 threads, 3 waves/SIMD regs, 150 KB LDS");
  return 0;
}
