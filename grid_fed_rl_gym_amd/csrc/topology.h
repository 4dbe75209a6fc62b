// topology.h -- host-side compilation of one feeder topology into the uniform tables the
// kernels read: Ybus CSR, unknown masks, elimination forest / sparse-LU schedule, per-bus
// device lists.  Pure host C++ (no HIP); runs once per gs_create.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gridstep.h"

struct HostTopology {
  int n = 0, m = 0, nnz = 0;
  int slack = 0;
  // Ybus CSR
  std::vector<int32_t> row_ptr, col;
  std::vector<double> G, B, Gd, Bd;
  std::vector<int32_t> ell_col, rem_ptr, rem_col;
  std::vector<double> ell_G, ell_B, rem_G, rem_B;
  std::vector<int32_t> th_free, vm_free, fixed_v;
  std::vector<double> v_set;
  // forest over active buses
  bool is_forest = false;
  int n_levels = 0, n_active = 0, max_level_width = 0;
  std::vector<int32_t> lvl_ptr, lvl_bus, parent, parent_pos, child_ptr, child_idx, lvl_pos;
  // FBS
  bool fbs_ok = false;
  std::string fbs_why;
  std::vector<int32_t> fbs_parent, fbs_parent_pos;
  // lines
  std::vector<int32_t> lfrom, lto;
  std::vector<double> lyr, lyi, lrating, lrating_inv;
  // sparse block LU
  bool has_lu = false;
  int lu_n_piv = 0, lu_n_slots = 0, lu_n_orig = 0;
  int64_t lu_n_pairs = 0;
  std::vector<int32_t> lu_piv_bus, lu_nb_ptr, lu_nb_bus, lu_nb_kj, lu_nb_jk;
  std::vector<int32_t> lu_pair_ptr, lu_pair_ik, lu_pair_kj, lu_pair_ij;
  std::vector<int32_t> lu_orig_slot, lu_orig_i, lu_orig_j, lu_orig_pos;
  // level of every pivot in the elimination DAG: 0 if no earlier pivot touched it, else 1 + the highest level among the
  // earlier pivots it was a neighbour of.  Pivots of one level are mutually non-adjacent in the filled graph, so their
  // eliminations commute and their updates never target each other's rows or columns.
  int lu_n_levels = 0;
  std::vector<int32_t> lu_piv_level;
  // dense LU unknown numbering
  int dn_N = 0;
  std::vector<int32_t> dn_th_idx, dn_vm_idx;
  // per-bus device lists
  std::vector<int32_t> bl_ptr, bl_idx, bg_ptr, bg_idx, bb_ptr, bb_idx;
  std::vector<double> load_base, load_q, gen_cap, gen_p0, gen_p1, gen_p2, bat_cap, bat_rating, bat_eff;
  std::vector<int32_t> gen_kind;
};

// Returns "" on success, else an error message.  `want_lu` forces building the sparse-LU
// schedule even for forests (GS_LINSOLVE_SPARSE_LU).
// schedule even for forests (GS_LINSOLVE_SPARSE_LU); `skip_lu` suppresses it (dense path: a
// near-complete graph would need O(n^3) schedule entries for nothing).
std::string gs_compile_topology(const gs_topology& t, int zero_z_mode, bool want_lu, bool skip_lu, HostTopology& out);
