// icache_probe.hip -- cycles per instruction of a straight-line loop body as its size grows past the instruction cache.
//   hipcc --offload-arch=gfx950 -O3 tools/icache_probe.hip -o build/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int COPIES>
__global__ void __launch_bounds__(512) body(double* out, long long* cyc, int reps) {
  double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < COPIES; ++i) {          // 16 instructions of 8 bytes per copy
      a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
      a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
      a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
      a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int COPIES> void run(double* out, long long* cyc, int threads) {
  const int reps = 40;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(body<COPIES>, dim3(256), dim3(threads), 0, 0, out, cyc, 2);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(body<COPIES>, dim3(256), dim3(threads), 0, 0, out, cyc, reps);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("loop body %4d KB, %4d threads per CU: wave 0 %6.2f counts per instruction, wall %6.2f ns per instruction per wave\n", COPIES * 128 / 1024, threads,
         (double)c / ((double)reps * COPIES * 16), ms * 1e6 / ((double)reps * COPIES * 16));
}

int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 1024 * sizeof(double)); (void)hipMalloc(&cyc, 8);
  for (int threads : {256, 512}) {
    run<64>(out, cyc, threads); run<128>(out, cyc, threads); run<192>(out, cyc, threads); run<256>(out, cyc, threads); run<320>(out, cyc, threads);
    run<384>(out, cyc, threads); run<512>(out, cyc, threads); run<768>(out, cyc, threads); run<1024>(out, cyc, threads);
  }
  return 0;
}
