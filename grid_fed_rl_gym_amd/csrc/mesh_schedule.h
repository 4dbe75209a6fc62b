// mesh_schedule.h -- host-side schedule of the meshed Newton-Raphson member of the second-generation step kernels
// (gs_k_step_nr_mesh2, kernels_flow2.hip): the sparse 2x2-block LU of a feeder with a few loops, laid out so that an
// instance's factorisation never leaves the chip.
//
// The linear solve the reference does densely (np.linalg.solve, environments/power_flow.py:186-190) is a block elimination in
// minimum-degree order.  What differs from the first-generation kernel (gs_k_step_nr_lu: blocks in slab rows, 14 x the
// algorithmic bytes through the fabric) is where a block lives and who touches it:
//
//   * PULL model.  Pivot k (remaining neighbours N(k)) gathers
//         D_k  = J_kk + sum of the C parts addressed to k          r_k = rhs_k + sum of the q parts
//         A_kj = J_kj + sum of the M blocks addressed to (k, j)     A_jk = J_jk + sum of those addressed to (j, k)     j in N(k)
//     forms  s_k = D_k^-1 r_k,  T_kj = D_k^-1 A_kj  and sends
//         CQ(k -> i) = (-A_ik T_ki, -A_ik s_k)     M(k -> (i, j)) = -A_ik T_kj  (i != j)            i, j in N(k)
//     to whoever eliminates the target first.  Back substitution: x_k = s_k - sum_j T_kj x_j.  No block is ever read, modified
//     and written by two parties; original Jacobian blocks are never stored (formed from the voltages where they are needed).
//   * A pivot of degree d takes a GROUP of max(d, 1) consecutive sub-groups of ONE wavefront row (8 sub-groups of 8 instances):
//     lane t handles neighbour j_t (both blocks, T, row j_t's messages); every lane of the group forms D_k, r_k, D_k^-1 and s_k for
//     itself (same instructions, same operands: nothing to exchange), lane 0 is the one that counts.  T stays in the registers of
//     the lane that formed it until the back substitution; everything else is a message in LDS.
//   * Messages ACCUMULATE: the producers of one target at different levels add into the same slot (a level barrier apart);
//     producers of the same level get accumulators of their own.  A target's pull list is then 1-4 entries whatever its
//     history, and the LDS footprint is the active submatrix, not the list of updates.
//   * Levels of the elimination DAG are the barriers of the kernel; a pivot may be delayed inside its window so that no
//     target gets more than `acc_cap` producers in one level.
//
// Everything the kernel needs per (wave, row, sub-group) is one GsMeshItem (192 bytes, pull lists inside) + a slice of the
// adjacency arrays; per (wave, row) four integers of `rowinfo`.  tests/test_mesh_schedule.py replays these tables in NumPy
// (no GPU) against a dense solve.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "topology.h"

#include "gs_internal.h"
typedef GsMeshItem MeshItem;             // (the record the kernel reads: gs_internal.h)
static_assert(sizeof(MeshItem) == 192, "GsMeshItem layout");
enum { MESH_F_PIVOT = GS_MESH_F_PIVOT, MESH_F_NBR = GS_MESH_F_NBR, MESH_F_SLACKPOS = GS_MESH_F_SLACKPOS, MESH_F_HV0_SHIFT = GS_MESH_F_HV0_SHIFT,
       MESH_F_T_SHIFT = GS_MESH_F_T_SHIFT, MESH_F_G_SHIFT = GS_MESH_F_G_SHIFT, MESH_F_RMW_SHIFT = GS_MESH_F_RMW_SHIFT };

struct MeshSchedule {
  bool ok = false; std::string why;
  int NW = 0, NI = 0, HV = 8, IW = 8;
  int n_levels = 0, n_rows = 0, max_rows_per_wave = 0, n_pivots = 0;
  int msg_units = 0;                     // 16-byte-per-lane units of the message region behind its header
  int n_messages = 0, n_accumulators = 0, max_degree = 0;
  // region layout (byte offsets from the region's start): ZERO (3 units, stays 0), DUMMY (3 units, write-only), then messages
  // and -- during the back substitution -- the x slots, one unit per bus
  int unit_bytes = 0, zero_off = 0, dummy_off = 0, body_off = 0, region_bytes = 0;
  std::vector<int32_t> rowinfo;          // [NW * NI][4]: level (-1: no row), g | ncq << 8 | nrw << 16 | ncl << 24, nadj, reserved
  std::vector<MeshItem> items;           // [NW * NI * 8]
  std::vector<int32_t> adj_off;          // per pivot lane: the Ybus row of its bus in CSR order (diagonal included), padded to nadj of its row
  std::vector<double> adj_y;             // (G, B) per entry
  // What the kernel reads: the same items packed into 16 words each (GS_MESH_W_*, gs_internal.h; read from global memory a row
  // ahead), and three small tables it keeps in LDS -- the off-diagonal Ybus entry of every connected pair of buses (entry n_pairs:
  // zero), the diagonal entry of every voltage slot ((0, -1) for the slots that are not buses: an identity-like diagonal block for
  // idle lanes), and every bus's neighbours as (pair | other bus << 16).  rowinfo[.][2] (nadj) counts off-diagonal entries here.
  int n_pairs = 0;
  std::vector<int32_t> packed;           // [NW * NI * 8][16]
  std::vector<double> ytab;              // [(n_pairs + 1) + (n + 3)][2]
  std::vector<int32_t> adj_ent;          // per bus contiguous
  std::vector<int32_t> rowinfo_packed;   // rowinfo with nadj = most off-diagonal neighbours of a pivot bus in the row
};

// region_base: LDS byte offset of the region (GsF2Tables::off_tile); slot_bytes: bytes of a voltage slot ((IW + 1) * 16).
// acc_cap: most accumulators per target (1 .. GS_MESH_ACC, the pull lists' capacity).  unit_budget: the message units the level
// assignment tries to stay below (what lets two workgroups share a CU); 0: levels as early as possible, no search.
void gs_mesh_schedule(const HostTopology& ht, int NW, int NI, int IW, int region_base, int slot_bytes, int acc_cap, int unit_budget, MeshSchedule& out);
