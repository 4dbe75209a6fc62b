// vgpr_probe.hip -- does the cost of an FP64 instruction depend on how many registers the wave holds / which it touches?
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N, int NT>
__global__ void __launch_bounds__(NT) body(const double* in, double* out, long long* cyc, int reps) {
  double x[N];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = in[i * NT + threadIdx.x];
  const double m = 1.0000001, c = 1e-9;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] = __builtin_fma(x[i], m, c); asm volatile("" : "+v"(x[i])); }
  }
  const long long t1 = __builtin_readcyclecounter();
  double acc = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) acc += x[i];
  out[blockIdx.x * NT + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int N, int NT> void run(const double* in, double* out, long long* cyc) {
  const int reps = 200;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((body<N, NT>), dim3(256), dim3(NT), 0, 0, in, out, cyc, 2);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((body<N, NT>), dim3(256), dim3(NT), 0, 0, in, out, cyc, reps);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%3d live doubles per thread, %4d threads per CU: wave 0 %6.2f counts per fma, wall %6.2f ns per fma per wave\n", N, NT, (double)c / ((double)reps * N), ms * 1e6 / ((double)reps * N));
}

int main() {
  double *in, *out; long long* cyc;
  (void)hipMalloc(&in, 128 * 1024 * sizeof(double)); (void)hipMalloc(&out, 256 * 1024 * sizeof(double)); (void)hipMalloc(&cyc, 8);
  (void)hipMemset(in, 0, 128 * 1024 * sizeof(double));
  run<16, 512>(in, out, cyc); run<60, 512>(in, out, cyc); run<100, 512>(in, out, cyc); run<120, 512>(in, out, cyc);
  run<16, 256>(in, out, cyc); run<120, 256>(in, out, cyc); run<200, 256>(in, out, cyc); run<240, 256>(in, out, cyc);
  run<16, 1024>(in, out, cyc); run<56, 1024>(in, out, cyc);
  return 0;
}
