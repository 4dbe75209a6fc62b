// permlane_probe.hip -- what v_permlane16_swap / v_permlane32_swap do on gfx950, as used by the cross-sub-group reductions of
// kernels_flow2.hip (f2_rows16 / f2_halves32): with both operands the same register x,
//   permlane32_swap -> (x of lane i mod 32, x of lane 32 + i mod 32)      on every lane
//   permlane16_swap -> (x of the EVEN 16-lane row of this lane's row pair, x of the ODD row)
// Checked against __shfl.  Build and run:  hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>

struct P2 { double a, b; };
__device__ __forceinline__ P2 swap32(double x) {
  const uint2 u = __builtin_bit_cast(uint2, x);
  auto lo = __builtin_amdgcn_permlane32_swap(u.x, u.x, false, false);
  auto hi = __builtin_amdgcn_permlane32_swap(u.y, u.y, false, false);
  return P2{__builtin_bit_cast(double, make_uint2(lo[0], hi[0])), __builtin_bit_cast(double, make_uint2(lo[1], hi[1]))};
}
__device__ __forceinline__ P2 swap16(double x) {
  const uint2 u = __builtin_bit_cast(uint2, x);
  auto lo = __builtin_amdgcn_permlane16_swap(u.x, u.x, false, false);
  auto hi = __builtin_amdgcn_permlane16_swap(u.y, u.y, false, false);
  return P2{__builtin_bit_cast(double, make_uint2(lo[0], hi[0])), __builtin_bit_cast(double, make_uint2(lo[1], hi[1]))};
}
__global__ void probe(const double* in, double* out) {
  const int lane = threadIdx.x;
  const double x = in[lane];
  const P2 a = swap16(x), b = swap32(x);
  out[lane] = a.a; out[64 + lane] = a.b; out[128 + lane] = b.a; out[192 + lane] = b.b;
  out[256 + lane] = __shfl(x, (lane & 32) | (lane & 15));          // even row of the pair
  out[320 + lane] = __shfl(x, (lane & 32) | 16 | (lane & 15));     // odd row of the pair
  out[384 + lane] = __shfl(x, lane & 31);
  out[448 + lane] = __shfl(x, 32 | (lane & 31));
}
int main() {
  double h[64], o[512]; double *di, *dout;
  for (int i = 0; i < 64; ++i) h[i] = 1000.0 + i + 1.0 / (3 + i);
  hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(di, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (o[i] != o[256 + i]) ++bad;
  printf("permlane16_swap / permlane32_swap vs __shfl: %d mismatches of 256\n", bad);
  if (bad) for (int i = 0; i < 64; i += 5) printf("lane %2d: swap16 (%.0f, %.0f) want (%.0f, %.0f); swap32 (%.0f, %.0f) want (%.0f, %.0f)\n", i, o[i], o[64 + i], o[256 + i], o[320 + i], o[128 + i], o[192 + i], o[384 + i], o[448 + i]);
  return bad ? 1 : 0;
}
