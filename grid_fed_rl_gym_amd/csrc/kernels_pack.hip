// kernels_pack.hip -- layout changes at the boundary, staged through LDS so that BOTH sides of
// every copy are coalesced: the C ABI speaks batch-major [B][C] (the reference's per-env
// vectors stacked: obs rows, P_spec rows), the kernels speak batch-innermost slab rows
// [row][64 lanes].  These are the HBM-streaming kernels of the step: obs packing moves
// B * obs_dim * 8 bytes in and out per step (grid_env.py:753-783 is the column order).
#include <hip/hip_runtime.h>

#include "gs_internal.h"

#define TILE_C 64
#define TILE_PAD 65

// out[b][c] = src[c] >= 0 ? slab row src[c] of instance b : cst[-src[c]-1]
// grid = (groups, ceil(C / 64)), block = 256 (4 waves).
extern "C" __global__ void __launch_bounds__(256)
gs_k_pack(const int32_t* __restrict__ src, const double* __restrict__ cst, int C, int rows_total,
          const double* __restrict__ slab, double* __restrict__ out, int B) {
  __shared__ double tile[TILE_C * TILE_PAD];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = blockIdx.x;
  const int c0 = blockIdx.y * TILE_C;
  const double* S = slab + (size_t)g * rows_total * GS_LANES;
  // load: one slab row (64 lanes, 512 B contiguous) per wave-instruction
  for (int cc = wave; cc < TILE_C; cc += 4) {
    const int c = c0 + cc;
    if (c < C) {
      const int s = src[c];
      tile[cc * TILE_PAD + lane] = (s >= 0) ? S[GS_ELEM(s, lane)] : cst[-s - 1];
    }
  }
  __syncthreads();
  // store: 64 consecutive columns of one instance (512 B contiguous) per wave-instruction
  const int c = c0 + lane;
  for (int r = wave; r < GS_LANES; r += 4) {
    const int b = g * GS_LANES + r;
    if (b < B && c < C) out[(size_t)b * C + c] = tile[lane * TILE_PAD + r];
  }
}

// slab row dst[c] of instance b = in[b][c], c < C, rows of `in` being `stride` doubles apart; same tiling, opposite direction.
extern "C" __global__ void __launch_bounds__(256)
gs_k_unpack(const int32_t* __restrict__ dst, int C, int rows_total, double* __restrict__ slab,
            const double* __restrict__ in, int B, int stride) {
  __shared__ double tile[TILE_C * TILE_PAD];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = blockIdx.x;
  const int c0 = blockIdx.y * TILE_C;
  double* S = slab + (size_t)g * rows_total * GS_LANES;
  const int c = c0 + lane;
  for (int r = wave; r < GS_LANES; r += 4) {
    const int b = g * GS_LANES + r;
    tile[lane * TILE_PAD + r] = (b < B && c < C) ? in[(size_t)b * stride + c] : 0.0;
  }
  __syncthreads();
  for (int cc = wave; cc < TILE_C; cc += 4) {
    const int cx = c0 + cc;
    if (cx < C) S[GS_ELEM(dst[cx], lane)] = tile[cc * TILE_PAD + lane];
  }
}

// Fill `count` slab rows row0, row0 + stride, ... with a constant (Q_spec = 0 when the caller passes NULL).
extern "C" __global__ void __launch_bounds__(64)
gs_k_fill_rows(int row0, int stride, int count, int rows_total, double* __restrict__ slab, double value) {
  double* S = slab + (size_t)blockIdx.x * rows_total * GS_LANES;
  for (int r = 0; r < count; ++r) S[GS_ELEM(row0 + stride * r, threadIdx.x)] = value;
}

// Per-instance scalars -> typed contiguous [B] arrays (already lane-contiguous in the slab).
//   f64 block: nf arrays of Bp doubles, i32 block: ni arrays, u8 block: nu arrays; each array
//   takes its values from one slab row (rows listed in rf / ri / ru).
extern "C" __global__ void __launch_bounds__(64)
gs_k_scalars(const int32_t* __restrict__ rf, int nf, const int32_t* __restrict__ ri, int ni,
             const int32_t* __restrict__ ru, int nu, int rows_total, const double* __restrict__ slab,
             double* __restrict__ of, int32_t* __restrict__ oi, uint8_t* __restrict__ ou, int Bp, uint32_t* __restrict__ ov4, int vf0) {
  const int b = blockIdx.x * GS_LANES + threadIdx.x;
  const double* S = slab + (size_t)blockIdx.x * rows_total * GS_LANES;
  const int ln = threadIdx.x;
  for (int k = 0; k < nf; ++k) of[(size_t)k * Bp + b] = S[GS_ELEM(rf[k], ln)];
  for (int k = 0; k < ni; ++k) oi[(size_t)k * Bp + b] = (int32_t)S[GS_ELEM(ri[k], ln)];
  uint32_t v4 = 0;
  for (int k = 0; k < nu; ++k) {
    const uint32_t f = (S[GS_ELEM(ru[k], ln)] != 0.0) ? 1u : 0u;
    ou[(size_t)k * Bp + b] = (uint8_t)f;
    if (k >= vf0 && k < vf0 + 4) v4 |= f << (8 * (k - vf0));
  }
  if (ov4) ov4[b] = v4;      // the four flag arrays ru[vf0 .. vf0 + 3] once more, as the bytes of one word per instance ([B][4] on the host)
}

// Observation blocks with / without their block of per-instance constants (columns [skip0, skip1): the static load
// powers, 27 % of an IEEE-123 observation) -- what the RCCL all-gather moves over xGMI is the compact form.
//   compact:  dst[r][j] = src[r][j < skip0 ? j : j + gap],  j < D - gap
//   expand :  dst[r][j < skip0 ? j : j + gap] = src[r][j]   (the constants of dst were written once, at gs_comm_init)
// One thread per (row, compact column), columns fastest: both sides are contiguous runs of skip0 and D - skip1 doubles.
extern "C" __global__ void __launch_bounds__(256)
gs_k_obs_compact(const double* __restrict__ src, double* __restrict__ dst, long long rows, int D, int skip0, int skip1, int expand) {
  const int gap = skip1 - skip0, nd = D - gap;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * nd) return;
  const long long r = idx / nd;
  const int j = (int)(idx - r * nd), c = j < skip0 ? j : j + gap;
  if (expand) dst[r * D + c] = src[idx];
  else dst[idx] = src[r * D + c];
}

// dst[k] = (float)src[k], two entries per thread (the observation block for a caller that asked for float32: gs_step_f32)
extern "C" __global__ void __launch_bounds__(256)
gs_k_obs_to_f32(const double* __restrict__ src, float* __restrict__ dst, long long n) {
  const long long k = 2 * ((long long)blockIdx.x * blockDim.x + threadIdx.x);
  if (k + 1 < n) {
    typedef double gs_d2 __attribute__((ext_vector_type(2)));
    typedef float gs_f2 __attribute__((ext_vector_type(2)));
    const gs_d2 v = *(const gs_d2*)(src + k);
    gs_f2 o; o.x = (float)v.x; o.y = (float)v.y;
    __builtin_nontemporal_store(o, (gs_f2*)(dst + k));
  } else if (k < n) {
    dst[k] = (float)src[k];
  }
}

// out[q] = row (row0 + q) of lane `lane` of group 0, q < count (gs_create: the handle's flat-start LU blocks, read off the
// rows one ordinary factorisation left there)
extern "C" __global__ void __launch_bounds__(256)
gs_k_gather_lane(int row0, int count, int lane, const double* __restrict__ slab, double* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < count) out[q] = slab[GS_ELEM(row0 + q, lane)];
}
