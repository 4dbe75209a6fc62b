// Stand-alone reproducer of the store-data hazard behind GsPairRef::put (grid_fed_rl_gym_amd/csrc/gs_internal.h):
// buffer_store_dwordx4 with the row offset in an SGPR (soffset) and `offen`, its four data registers overwritten by
// VALU moves N wait states later.  Counts the 16-byte slots that did not receive the value.
//   hipcc -O2 --offload-arch=gfx950 tools/store_data_hazard.hip -o /tmp/hazard && /tmp/hazard
// MI355X, ROCm 7.2 (round 1):   wait states 0: 109632 of 16777216 slots wrong, all of them lanes 12-15 of every 16
//                               wait states 1, 2, 4: 0 wrong
// The compiler's hazard recogniser inserts that wait state for 16-byte stores EXCEPT when soffset is a register.
// (The same experiment with buffer_store_dwordx2, 8 bytes per lane: 0 of 16777216 wrong -- only the wide store is affected.)
// Round 2, the other store forms with ZERO wait states (k2 below; same box, ROCm 7.2):
//   buffer_store_dwordx4, soffset in an SGPR    111200 of 16777216 slots wrong, lanes 12-15 of every 16      <- the one the compiler leaves unprotected
//   global_store_dwordx4 ... saddr              4004048 of 16777216 slots wrong (the compiler pads this form itself: never emitted bare)
//   buffer_store_dwordx2, soffset in an SGPR    0 wrong
//   global_store_dwordx2 ... saddr              0 wrong
// tools/check_store_hazard.py checks every 12- / 16-byte buffer and global store of the shipped library for a VALU write of
// its data registers in the next instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int NOPS>
__global__ void __launch_bounds__(1024) k(unsigned* buf, unsigned long long bytes_per_group, int iters, const double* pressure, double* sink) {
  unsigned* g = buf + (size_t)blockIdx.x * (bytes_per_group / 4);
  const unsigned long long base = (unsigned long long)g;
  u4 rsrc; rsrc.x = (unsigned)base; rsrc.y = (unsigned)(base >> 32); rsrc.z = (unsigned)bytes_per_group; rsrc.w = 0x00020000u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  const unsigned voff = (unsigned)lane << 4;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    const int row = it * W + wave;                       // every wave its own 1 KB row per iteration
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(row << 10);
    const unsigned a = 0x3ff00000u + it, b = 0x11111111u, c = 0x22222222u, d = 0x33333333u;   // never zero
    acc += pressure[(size_t)(threadIdx.x + it * 1024) & 0xfffff];                              // memory traffic around the store
    if (NOPS == 0)
      asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\ts_nop 7\n\t"
                   "buffer_store_dwordx4 v[40:43], %4, %5, %6 offen\n\t"
                   "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0"
                   :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rsrc), "s"(soff) : "v40", "v41", "v42", "v43", "memory");
    else if (NOPS == 1)
      asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\ts_nop 7\n\t"
                   "buffer_store_dwordx4 v[40:43], %4, %5, %6 offen\n\ts_nop 0\n\t"
                   "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0"
                   :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rsrc), "s"(soff) : "v40", "v41", "v42", "v43", "memory");
    else if (NOPS == 2)
      asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\ts_nop 7\n\t"
                   "buffer_store_dwordx4 v[40:43], %4, %5, %6 offen\n\ts_nop 1\n\t"
                   "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0"
                   :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rsrc), "s"(soff) : "v40", "v41", "v42", "v43", "memory");
    else
      asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\ts_nop 7\n\t"
                   "buffer_store_dwordx4 v[40:43], %4, %5, %6 offen\n\ts_nop 3\n\t"
                   "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0"
                   :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rsrc), "s"(soff) : "v40", "v41", "v42", "v43", "memory");
  }
  if (acc == 12345.678) sink[0] = acc;
}

template <int NOPS> void run(unsigned* d, size_t bytes_per_group, int groups, int iters, const double* pr, double* sink, std::vector<unsigned>& h) {
  (void)hipMemset(d, 0xff, bytes_per_group * groups);
  hipLaunchKernelGGL(k<NOPS>, dim3(groups), dim3(1024), 0, 0, d, (unsigned long long)bytes_per_group, iters, pr, sink);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h.data(), d, bytes_per_group * groups, hipMemcpyDeviceToHost);
  long bad = 0, total = 0; long by_lane[16] = {0};
  for (int g = 0; g < groups; ++g)
    for (int row = 0; row < iters * 16; ++row)
      for (int lane = 0; lane < 64; ++lane) {
        const unsigned* p = h.data() + (size_t)g * (bytes_per_group / 4) + (size_t)row * 256 + lane * 4;
        const unsigned a = 0x3ff00000u + row / 16;
        ++total;
        if (p[0] != a || p[1] != 0x11111111u || p[2] != 0x22222222u || p[3] != 0x33333333u) { ++bad; ++by_lane[lane & 15]; }
      }
  printf("wait states %d: %ld of %ld slots wrong; by lane mod 16:", NOPS == 3 ? 4 : NOPS, bad, total);
  for (int l = 0; l < 16; ++l) printf(" %ld", by_lane[l]);
  printf("\n");
}

// The other store forms, each with ZERO wait states between the store and the overwrite of its data registers
// (recorded once, MI355X / ROCm 7.2, round 2 -- see the header of this file for the results):
//   FORM 0  global_store_dwordx4 v, v[40:43], s[base:base+1]     (saddr form: what `*(double2*)p = v` compiles to when
//                                                                 the base is uniform; the compiler pads this one itself)
//   FORM 1  buffer_store_dwordx2 v[40:41], voff, rsrc, soff offen (8-byte row store, GsRowRef::put)
//   FORM 2  global_store_dwordx2 v, v[40:41], s[base:base+1]
template <int FORM>
__global__ void __launch_bounds__(1024) k2(unsigned* buf, unsigned long long bytes_per_group, int iters, const double* pressure, double* sink) {
  unsigned* g = buf + (size_t)blockIdx.x * (bytes_per_group / 4);
  const unsigned long long base = (unsigned long long)g;
  u4 rsrc; rsrc.x = (unsigned)base; rsrc.y = (unsigned)(base >> 32); rsrc.z = (unsigned)bytes_per_group; rsrc.w = 0x00020000u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    const int row = it * W + wave;
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(row << 10);
    const unsigned long long rowbase = base + soff;
    const unsigned voff = (unsigned)lane << 4;
    const unsigned a = 0x3ff00000u + it, b = 0x11111111u, c = 0x22222222u, d = 0x33333333u;
    acc += pressure[(size_t)(threadIdx.x + it * 1024) & 0xfffff];
    if (FORM == 0)
      asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\ts_nop 7\n\t"
                   "global_store_dwordx4 %4, v[40:43], %5\n\t"
                   "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0"
                   :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rowbase) : "v40", "v41", "v42", "v43", "memory");
    else if (FORM == 1)
      asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\ts_nop 7\n\t"
                   "buffer_store_dwordx2 v[40:41], %2, %3, %4 offen\n\t"
                   "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0"
                   :: "v"(a), "v"(b), "v"(voff), "s"(rsrc), "s"(soff) : "v40", "v41", "memory");
    else
      asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\ts_nop 7\n\t"
                   "global_store_dwordx2 %2, v[40:41], %3\n\t"
                   "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0"
                   :: "v"(a), "v"(b), "v"(voff), "s"(rowbase) : "v40", "v41", "memory");
  }
  if (acc == 12345.678) sink[0] = acc;
}

template <int FORM> void run2(const char* what, unsigned* d, size_t bytes_per_group, int groups, int iters, const double* pr, double* sink, std::vector<unsigned>& h) {
  (void)hipMemset(d, 0xff, bytes_per_group * groups);
  hipLaunchKernelGGL(k2<FORM>, dim3(groups), dim3(1024), 0, 0, d, (unsigned long long)bytes_per_group, iters, pr, sink);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h.data(), d, bytes_per_group * groups, hipMemcpyDeviceToHost);
  long bad = 0, total = 0;
  const int words = FORM == 0 ? 4 : 2;
  for (int g = 0; g < groups; ++g)
    for (int row = 0; row < iters * 16; ++row)
      for (int lane = 0; lane < 64; ++lane) {
        const unsigned* p = h.data() + (size_t)g * (bytes_per_group / 4) + (size_t)row * 256 + lane * 4;
        const unsigned want[4] = {0x3ff00000u + row / 16, 0x11111111u, 0x22222222u, 0x33333333u};
        ++total;
        bool ok = true;
        for (int w = 0; w < words; ++w) ok = ok && p[w] == want[w];
        if (!ok) ++bad;
      }
  printf("form %s, 0 wait states: %ld of %ld slots wrong\n", what, bad, total);
}

int main() {
  const int groups = 256, iters = 64;
  const size_t bytes_per_group = (size_t)iters * 16 * 1024;
  unsigned* d; (void)hipMalloc(&d, bytes_per_group * groups);
  double *pr, *sink; (void)hipMalloc(&pr, (1 << 20) * 8); (void)hipMemset(pr, 0, (1 << 20) * 8); (void)hipMalloc(&sink, 8);
  std::vector<unsigned> h(bytes_per_group * groups / 4);
  run<0>(d, bytes_per_group, groups, iters, pr, sink, h);
  run<1>(d, bytes_per_group, groups, iters, pr, sink, h);
  run<2>(d, bytes_per_group, groups, iters, pr, sink, h);
  run<3>(d, bytes_per_group, groups, iters, pr, sink, h);
  run2<0>("global_store_dwordx4_saddr", d, bytes_per_group, groups, iters, pr, sink, h);
  run2<1>("buffer_store_dwordx2_soffset", d, bytes_per_group, groups, iters, pr, sink, h);
  run2<2>("global_store_dwordx2_saddr", d, bytes_per_group, groups, iters, pr, sink, h);
  return 0;
}
