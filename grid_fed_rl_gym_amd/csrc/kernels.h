// kernels.h -- host-visible declarations of the HIP kernels (defined in kernels_*.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "gs_internal.h"

#define GS_DECLARE_KERNELS(name)                                                                               \
  __global__ void gs_k_##name(GsTables T, GsRows R, GsSolveCfg C, double* __restrict__ slab, int B);           \
  __global__ void gs_k_step_##name(GsTables T, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab,  \
                                   int B, const double* __restrict__ actions, double total_load, GsPackArgs PA,        \
                                   GsFusedChecks FC);                                                          \
  __global__ void gs_k_stepc_##name(GsTables T, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, \
                                    int B, const double* __restrict__ actions, double total_load, GsPackArgs PA,       \
                                    GsFusedChecks FC);

extern "C" {
GS_DECLARE_KERNELS(nr_tree)
GS_DECLARE_KERNELS(nr_tree_lds)
GS_DECLARE_KERNELS(nr_lu)
GS_DECLARE_KERNELS(nr_dense)
GS_DECLARE_KERNELS(fbs)
GS_DECLARE_KERNELS(fbs_lds)
GS_DECLARE_KERNELS(fbs_flow)
__global__ void gs_k_step_fbs_flow2(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_step_fbs_flow2h(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_step_fbs_flow2x(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_stepc_fbs_flow2x(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_stepc_fbs_flow2h(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_stepc_fbs_flow2s(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_step_fbs_flow2s(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_stepc_nr_flow2s(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_step_nr_flow2s(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);

__global__ void gs_k_stepc_fbs_flow2(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                     const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_step_nr_flow2(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                   const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_stepc_nr_flow2(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_step_nr_mesh2(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                   const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_stepc_nr_mesh2(GsTables T, GsF2Tables F, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC, GsRolloutStep RS);
__global__ void gs_k_pre_nr_dmfma(GsTables T, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                  const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC);
__global__ void gs_k_post_nr_dmfma(GsTables T, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                   const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC);
__global__ void gs_k_postc_nr_dmfma(GsTables T, GsRows R, GsSolveCfg C, GsEnvCfg E, double* __restrict__ slab, int B,
                                    const double* __restrict__ actions, double total_load, GsPackArgs PA, GsFusedChecks FC);
__global__ void gs_k_posts_nr_dmfma(GsTables T, GsRows R, GsSolveCfg C, double* __restrict__ slab, int B);
__global__ void gs_k_nr_dense_mfma(GsDenseArgs A, double* __restrict__ slab, int B);
__global__ void gs_k_nr_dense_mfma2(GsDenseArgs A, double* __restrict__ slab, int B);
__global__ void gs_k_nr_sparse_lds(GsSparseArgs A, double* __restrict__ slab, int B);
__global__ void gs_k_env_reset(GsTables T, GsRows R, GsEnvCfg E, double* __restrict__ slab, int B,
                               const uint64_t* __restrict__ seeds, const uint8_t* __restrict__ mask);
__global__ void gs_k_polar_to_rect(GsTables T, GsRows R, double* __restrict__ slab, int B);
__global__ void gs_k_rollout_actions(double* __restrict__ act, int T, int B, int A, uint64_t seed, int64_t first_instance, uint32_t t0);
__global__ void gs_k_fill_const_columns(double* __restrict__ out, long long rows, int obs_dim, int skip0, int skip1,
                                        const int32_t* __restrict__ map, const double* __restrict__ cst);
__global__ void gs_k_rollout_post(GsTables T, GsRows R, GsEnvCfg E, double* __restrict__ slab, GsRolloutPostArgs A);
__global__ void gs_k_pack(const int32_t* __restrict__ src, const double* __restrict__ cst, int C, int rows_total,
                          const double* __restrict__ slab, double* __restrict__ out, int B);
__global__ void gs_k_unpack(const int32_t* __restrict__ dst, int C, int rows_total, double* __restrict__ slab,
                            const double* __restrict__ in, int B, int stride);
__global__ void gs_k_obs_compact(const double* __restrict__ src, double* __restrict__ dst, long long rows, int D, int skip0, int skip1, int expand);
__global__ void gs_k_obs_to_f32(const double* __restrict__ src, float* __restrict__ dst, long long n);
__global__ void gs_k_gather_lane(int row0, int count, int lane, const double* __restrict__ slab, double* __restrict__ out);
__global__ void gs_k_fill_rows(int row0, int stride, int count, int rows_total, double* __restrict__ slab, double value);
__global__ void gs_k_scalars(const int32_t* __restrict__ rf, int nf, const int32_t* __restrict__ ri, int ni,
                             const int32_t* __restrict__ ru, int nu, int rows_total, const double* __restrict__ slab,
                             double* __restrict__ of, int32_t* __restrict__ oi, uint8_t* __restrict__ ou, int Bp, uint32_t* __restrict__ ov4, int vf0);
__global__ void gs_k_checks(GsChecksCfg C, const double* __restrict__ slab, const double* __restrict__ freq_override,
                            double* __restrict__ prev, int32_t* __restrict__ state, int32_t* __restrict__ out_i,
                            double* __restrict__ out_f, uint8_t* __restrict__ bus_mask, uint8_t* __restrict__ line_mask, int B, int Bp);
__global__ void gs_k_fallback_linear(GsTables T, GsRows R, GsFallbackArgs A, double* __restrict__ slab, int B);
__global__ void gs_k_checks_reset(int32_t* __restrict__ state, const uint8_t* __restrict__ mask, int B, int Bp);
}
