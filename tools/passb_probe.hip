// passb_probe.hip -- the mismatch loop of gs3_k_resident in isolation: cycles per position for a wave, by what is in the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/passb_probe.hip -o build/passb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <math.h>

__device__ __forceinline__ double r3_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  return x;
}
#define PIN(x, y) asm volatile("" : "+v"(x), "+v"(y))

// VAR bit 0: rcp; bit 1: min/max mismatch; bit 2: root mask; bit 3: pins + fences
template <int K, int VAR, int NT>
__global__ void __launch_bounds__(NT) passb(const double* in, double* out, long long* cyc, int reps) {
  double ar[K], ai[K], br[K], bi[K]; int pk[K];
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; ++k) { ar[k] = in[k * NT + tid]; ai[k] = in[(K + k) * NT + tid]; br[k] = 1.0 + ar[k]; bi[k] = ai[k]; pk[k] = (int)(ar[k] * 1e6); }
  double lmax = 0.0, psum = 0.0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("" : "+v"(pk[k]));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double2 s = make_double2(-1e-4, -3e-5);
      if ((VAR & 4) && ((pk[k] >> 28) & 1)) s = make_double2(0.0, 0.0);
      const double wr = br[k], wi = bi[k];
      const double pc = -(wr * ar[k] + wi * ai[k]), qc = -(wi * ar[k] - wr * ai[k]);
      if (VAR & 2) {
        const double dP = fabs(s.x - pc), dQ = fabs(s.y - qc);
        lmax = fmax(lmax, fmax(dP < INFINITY ? dP : INFINITY, dQ < INFINITY ? dQ : INFINITY));
      } else lmax += pc * qc;
      psum += pc;
      const double rd = (VAR & 1) ? r3_rcp(wr * wr + wi * wi) : 2.0 - (wr * wr + wi * wi);
      ar[k] = -(s.x * wr + s.y * wi) * rd; ai[k] = -(s.x * wi - s.y * wr) * rd;
      if (VAR & 8) { PIN(ar[k], ai[k]); PIN(lmax, psum); if (k % 2 == 1) __builtin_amdgcn_sched_barrier(0); }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double acc = lmax + psum;
#pragma unroll
  for (int k = 0; k < K; ++k) acc += ar[k] + ai[k];
  out[blockIdx.x * NT + tid] = acc;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K, int VAR, int NT> void run(const char* name, const double* in, double* out, long long* cyc) {
  const int reps = 200;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((passb<K, VAR, NT>), dim3(256), dim3(NT), 0, 0, in, out, cyc, 2);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((passb<K, VAR, NT>), dim3(256), dim3(NT), 0, 0, in, out, cyc, reps);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("K=%2d threads=%4d %-34s: wave 0 %7.1f counts per position, wall %7.1f ns per position\n", K, NT, name, (double)c / (reps * K), ms * 1e6 / (reps * K));
}

int main() {
  double *in, *out; long long* cyc;
  (void)hipMalloc(&in, 80 * 1024 * sizeof(double)); (void)hipMalloc(&out, 256 * 1024 * sizeof(double)); (void)hipMalloc(&cyc, 8);
  static double h[80 * 1024];
  for (int i = 0; i < 80 * 1024; ++i) h[i] = 1e-4 * ((i * 7919) % 1000) / 1000.0;
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<19, 15, 512>("full (rcp, min/max, mask, pins)", in, out, cyc);
  run<19, 7, 512>("no pins / fences", in, out, cyc);
  run<19, 14, 512>("no rcp", in, out, cyc);
  run<19, 13, 512>("no min/max", in, out, cyc);
  run<19, 8, 512>("fma only + pins", in, out, cyc);
  run<19, 15, 256>("full, one wave per SIMD", in, out, cyc);
  run<37, 15, 256>("full, K = 37, one wave per SIMD", in, out, cyc);
  run<5, 15, 512>("full, K = 5", in, out, cyc);
  return 0;
}
