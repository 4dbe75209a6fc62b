// f64_rate_probe.hip -- issue cost of FP64 vector instructions on gfx950, and what s_memtime counts.
//   hipcc --offload-arch=gfx950 -O3 tools/f64_rate_probe.hip -o /tmp/f64_rate_probe && /tmp/f64_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(1024) probe(double* out, long long* cyc, int iters) {
  double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#define EIGHT(stmt) { double& x = a0; stmt } { double& x = a1; stmt } { double& x = a2; stmt } { double& x = a3; stmt } { double& x = a4; stmt } { double& x = a5; stmt } { double& x = a6; stmt } { double& x = a7; stmt }
    if (OP == 0) { EIGHT(x = __builtin_fma(x, m, c);) }
    if (OP == 1) { EIGHT(x = x * m;) }
    if (OP == 2) { EIGHT(x = x + c;) }
    if (OP == 3) { EIGHT(x = __builtin_amdgcn_rcp(x);) }
    if (OP == 4) { EIGHT(x = fmin(x, m);) }
    if (OP == 5) { EIGHT(asm volatile("v_mov_b64 %0, %0" : "+v"(x));) }
    if (OP == 6) { a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
                   a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); }
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(double)); hipMalloc(&cyc, 8);
  const char* names[] = {"v_fma_f64 x8 independent", "v_mul_f64 x8", "v_add_f64 x8", "v_rcp_f64 x8", "v_min_f64 x8", "v_mov_b64 x8", "v_fma_f64 x8 dependent chain"};
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {64, 256, 512, 1024}) {
    for (int op = 0; op < 7; ++op) {
      void (*k)(double*, long long*, int) = op == 0 ? probe<0> : op == 1 ? probe<1> : op == 2 ? probe<2> : op == 3 ? probe<3> : op == 4 ? probe<4> : op == 5 ? probe<5> : probe<6>;
      hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, 100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("threads/CU %4d (%d waves/SIMD) %-30s: %6.2f counts per instruction per wave, %7.3f ns per instruction (event), counter %.3f GHz\n", threads, threads / 256 ? threads / 256 : 1,
             names[op], (double)c / (8.0 * iters), ms * 1e6 / (8.0 * iters), c / (ms * 1e6));
    }
  }
  return 0;
}
