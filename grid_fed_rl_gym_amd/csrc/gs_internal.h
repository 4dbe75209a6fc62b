// gs_internal.h -- structures shared by the host side of libgridstep.so and its HIP kernels.
//
// Execution model: one *group* = 64 feeder instances = the 64 lanes of a
// wavefront; every lane runs the same topology-uniform program on its own instance, so
// control flow never diverges and every per-instance array is stored batch-innermost:
//
//     slab[group][row][lane]        (row stride 64 doubles = 512 B = one coalesced wave access)
//
// A workgroup is W (1..16) wavefronts that all own the SAME 64 instances and split the
// buses / lines / tree levels of the feeder between them (wave w takes items w, w+W, ...),
// meeting at workgroup barriers.  Topology tables are wave-uniform and are read through the
// scalar cache (s_load), never per lane.
#pragma once
#include <stdint.h>

// Switches that exist to MEASURE alternatives (kernel members AUTO never takes, table layouts, launch shapes) are read only by a
// library built with `make EXPERIMENTS=1` (-DGS_BUILD_EXPERIMENTS); the default build answers them with "not set" and does not
// carry the kernels only they can reach (gs_k_nr_sparse_lds, the 32-instance sweep member).  What stays readable in every build
// are the switches the parity tests compare builds of one handle with: GS_NO_FLOW2, GS_NO_FLOW, GS_NO_FLOW2_SMALL, GS_NO_SPLIT,
// GS_EAGER_ROWS, GS_NR_NO_FLAT, GS_LU_NO_FLAT, GS_DENSE_NO_FLAT, GS_NO_MESH2, GS3_NO_RESIDENT, GS3_DENSE_MUTUAL, GS_WAVES, and
// the two that arm the in-kernel clocks bench.py reads (GS_STAMP_WAVE, GS_STAMP_BLOCK_TIMES).
#if defined(GS_BUILD_EXPERIMENTS)
#define GS_EXPERIMENT_ENV(name) getenv(name)
#else
#define GS_EXPERIMENT_ENV(name) ((const char*)nullptr)
#endif

#define GS_LANES 64
// Slab layout inside a 64-instance group: rows come in PAIRS that share 1 KB, lane-interleaved --
//   double index of (row, lane) = (row / 2) * 128 + lane * 2 + (row % 2)
// so two logically adjacent rows (e and f of a bus, P and Q, J re and im, |V| and angle ...) are ONE 16-byte access per
// lane.  The kernels are bound by the number of vector memory instructions, not by their bytes (splitting every 8-byte
// load in two: +23 % / +38 % step time), which is what the pairing halves.  The row count of a group is kept even.
#define GS_ELEM(row, lane) ((((size_t)(row) >> 1) << 7) + ((size_t)(lane) << 1) + ((size_t)(row) & 1))
#define GS_MAX_WAVES 16
#define GS_ELL_K 4        // Ybus rows are stored in chunks of 4 entries (padded with exact zeros)

// pointer into the constant address space: forces scalar (s_load) access for uniform tables
#if defined(__HIP_DEVICE_COMPILE__)
#define GS_CONST __attribute__((address_space(4)))
#else
#define GS_CONST
#endif

#if defined(__HIPCC__)
// A lane's view of its group's slab rows, through a buffer descriptor.  Row r of the lane is at
// group base + r * 512 B + lane * 8 B; with the descriptor holding the group base, a wave-uniform row index goes into
// the instruction's scalar offset and the lane offset is one constant VGPR:
//     buffer_load_dwordx2 v, v_lane8, s[rsrc], s_rowoff offen
// so an access costs no vector ALU work (a 64-bit pointer per lane costs a v_lshl_add_u64 each time).  Indexed like the
// lane pointer it replaces, S[row * GS_LANES]; rows that differ between lanes (the dense solver's pivots) use
// S.lane_row(row * GS_LANES).  Out-of-range offsets read 0 / are dropped by the hardware bounds check.
//
// Row loads carry sc0 (workgroup scope): rows are how the waves of a group hand data to each other across a barrier.
// (Introduced while hunting "stale 64-byte sectors" that turned out to be the store-data hazard handled in
// GsPairRef::put below; costs nothing measurable, kept.)
#ifndef GS_LOAD_AUX
#define GS_LOAD_AUX 1
#endif
typedef unsigned int gs_u32x2 __attribute__((ext_vector_type(2)));
struct GsRowRef {
  __amdgpu_buffer_rsrc_t r;
  unsigned voff;
  int soff;
  __device__ __forceinline__ double get() const { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, GS_LOAD_AUX)); }
  __device__ __forceinline__ void put(double v) const { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(gs_u32x2, v), r, voff, soff, 0); }
  __device__ __forceinline__ operator double() const { return get(); }
  __device__ __forceinline__ double operator=(double v) const { put(v); return v; }
  __device__ __forceinline__ double operator=(const GsRowRef& o) const { const double v = o.get(); put(v); return v; }
  __device__ __forceinline__ double operator+=(double v) const { const double x = get() + v; put(x); return x; }
  __device__ __forceinline__ double operator-=(double v) const { const double x = get() - v; put(x); return x; }
};
typedef unsigned int gs_u32x4 __attribute__((ext_vector_type(4)));
struct GsPairRef {          // an (even row, odd row) pair of one lane: 16 bytes, one instruction
  __amdgpu_buffer_rsrc_t r;
  unsigned voff;
  int soff;
  __device__ __forceinline__ double2 get() const { return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, GS_LOAD_AUX)); }
  // A 16-byte store reads its data registers over several cycles after it issues.  With the row offset in an SGPR
  // (soffset) the compiler's hazard recogniser assumes the hardware interlocks (the documented exemption) and lets the
  // next VALU instruction overwrite them at once -- measured on gfx950 it does not: lanes 12-15 of every 16 then store
  // the NEW register contents (tools/store_data_hazard.hip).  So the STORE carries its row offset in the vector offset
  // (one v_add) and an immediate soffset: that is the form for which the compiler itself keeps the required wait state,
  // and every other hazard of a real VMEM instruction too.  (Two earlier forms were both wrong: a separate `s_nop` asm
  // behind the builtin let the compiler re-materialise operands in between; store + nop as one asm statement hid the
  // instruction from the hazard recogniser altogether -- with SGPRs spilled to VGPR lanes a v_readlane of the
  // descriptor landed right in front of the store, inside the 5 wait states a VALU-written SGPR needs before a VMEM
  // instruction reads it, and the rows of one kernel variant were silently not written.)
  // tools/check_store_hazard.py + tests/test_store_hazard_static.py check the emitted code of the whole library.
  __device__ __forceinline__ void put(double2 v) const {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gs_u32x4, v), r, voff + (unsigned)soff, 0, 0);
  }
  __device__ __forceinline__ operator double2() const { return get(); }
  __device__ __forceinline__ void operator=(double2 v) const { put(v); }
};
struct GsLaneRows {
  __amdgpu_buffer_rsrc_t r;
  char* g;                   // group base (uniform)
  unsigned lane16;
  // idx = row * GS_LANES, as for the lane pointer this replaces
  __device__ __forceinline__ GsRowRef operator[](size_t idx) const {                      // uniform row
    const unsigned row = (unsigned)(idx >> 6);
    return GsRowRef{r, lane16, (int)(((row >> 1) << 10) | ((row & 1) << 3))};
  }
  __device__ __forceinline__ GsPairRef pair(size_t idx) const {                           // uniform EVEN row and its odd partner
    return GsPairRef{r, lane16, (int)((unsigned)(idx >> 7) << 10)};
  }
  __device__ __forceinline__ GsRowRef lane_row(size_t idx) const {                        // per-lane row
    const unsigned row = (unsigned)(idx >> 6);
    return GsRowRef{r, lane16 + (((row >> 1) << 10) | ((row & 1) << 3)), 0};
  }
};
__device__ __forceinline__ GsLaneRows gs_lane_rows(double* slab, int group, int rows_total, int lane) {
  char* g = (char*)(slab + (size_t)group * rows_total * GS_LANES);
  return GsLaneRows{__builtin_amdgcn_make_buffer_rsrc(g, 0, rows_total * GS_LANES * (int)sizeof(double), 0x00020000), g, (unsigned)lane << 4};
}
#endif

// gs_k_fallback_linear (kernels_fallback.hip)
struct GsFallbackArgs {
  const double* load_w; const double* gen_w;          // [B][n] batch-major, or NULL with env_mode
  const double* tot_load; const double* tot_gen;      // [B] or NULL
  const int32_t* load_order; const int32_t* gen_order;   // buses in the order the reference's dicts are filled (env_mode)
  const double* line_x;                                // [m] line reactance
  const uint8_t* mask;                                 // [B] or NULL = where CONV == 0
  int32_t* applied;                                    // [B] out
  int32_t n_load_order, n_gen_order, env_mode, pad;
};

// One forest work item (a bus, in the order ONE wave meets it in the sweeps) with everything the
// sweep needs that does not depend on the instance: read as one contiguous scalar load, and the
// next record of the wave is the next 96 bytes (prefetched one item ahead).
#define GS_ITEM_CHILDREN 8
struct GsItemRec {
  int32_t bus, parent, slot, parent_slot;     // message slots already include the level parity
  int32_t n_children, flags, level, ovf0;     // flags: bit0 th_i, bit1 vf_i, bit2 th_p, bit3 vf_p; ovf0: first overflow child
  int32_t child_slot[GS_ITEM_CHILDREN];       // slots of the first 8 children (more: ovf_slot[ovf0 ...])
  double g, b, gd, bd;                        // Y(bus, parent) and Y(bus, bus)
};

// One bus of the mismatch pass, in the order ONE wave meets it: its Ybus row in ELL(8) form.
struct GsBusRec {
  int32_t bus, flags, pad0, pad1;             // flags: bit0 th_free, bit1 vm_free, bit2 row continues in the next record,
                                              //        bit3 this record continues the previous one
  int32_t col[GS_ELL_K];                      // neighbours (self included), padded with self
  double G[GS_ELL_K], B[GS_ELL_K];            // padded with exact zeros
};

// Devices attached to one bus, for the injection build (order = the reference's accumulation
// order, grid_env.py:689-718).  Buses with more than 2 devices of a kind set `generic` and go
// through the per-bus lists instead.
struct GsInjRec {
  int32_t bus, generic, nl, ng, nb, l0, l1, g0, g1, b0, b1, pad;
};

// Topology-uniform tables (device pointers; built once per handle by topology.cpp).
struct GsTables {
  int32_t n, m, nnz, n_levels;
  int32_t n_loads, n_gens, n_bats, pad0;
  // Ybus CSR over buses (diagonal included; columns ascending)
  const int32_t* row_ptr;    // [n+1]
  const int32_t* col;        // [nnz]
  const double* G;           // [nnz]
  const double* Bv;          // [nnz]
  // the same rows in ELL form for the mismatch: entry k of row i at [i * GS_ELL_K + k], rows
  // shorter than GS_ELL_K padded with (col = i, G = B = 0); longer rows continue in rem_*
  const int32_t* ell_col;    // [n * GS_ELL_K]
  const double* ell_G;       // [n * GS_ELL_K]
  const double* ell_B;       // [n * GS_ELL_K]
  const int32_t* rem_ptr;    // [n+1]
  const int32_t* rem_col;
  const double* rem_G;
  const double* rem_B;
  const double* Gd;          // [n] diagonal
  const double* Bd;          // [n]
  // unknown masks: th_free = bus has a theta unknown / P equation (non-slack, power_flow.py:232);
  // vm_free = bus has a Vm unknown / Q equation (pq list, power_flow.py:135-136)
  const int32_t* th_free;    // [n]
  const int32_t* vm_free;    // [n]
  const double* v_set;       // [n]
  const int32_t* fixed_v;    // [n] 1 where the flat start uses v_set (slack / pv, power_flow.py:128-134)
  // elimination forest over the active (non-slack) buses, deepest level first
  const int32_t* lvl_ptr;    // [n_levels+1]
  const int32_t* lvl_bus;    // [n_active]
  const int32_t* parent;     // [n] parent bus in the forest, -1 for a root / inactive bus
  const int32_t* parent_pos; // [n] CSR position of entry (i, parent[i])
  const int32_t* child_ptr;  // [n+1]
  const int32_t* child_idx;  // [n_active - n_roots]
  const int32_t* lvl_pos;    // [n] position of bus i inside its level (message slot), -1 if inactive
  // per-wave work lists of the forest sweeps (built for the handle's waves_per_group)
  const GsItemRec* witems;   // records grouped by wave, bottom-up order inside a wave
  const int32_t* wl_ptr;     // [W+1]
  const int32_t* ovf_slot;   // child slots beyond the 8 kept in the record
  const GsInjRec* winj;      // injection records grouped by wave (wave w: buses w, w+W, ...)
  const int32_t* wi_ptr;     // [W+1]
  const GsBusRec* wbus;      // mismatch records grouped by wave (buses dealt to waves by row length, longest first)
  const int32_t* wb_ptr;     // [W+1]
  int32_t max_level_width, pad3;
  // FBS: tree rooted at the slack bus (levels exclude the slack itself); fbs_parent includes the slack
  const int32_t* fbs_parent;     // [n]
  const int32_t* fbs_parent_pos; // [n]
  // lines
  const int32_t* lfrom;      // [m]
  const int32_t* lto;        // [m]
  const double* lyr;         // [m] series admittance, real
  const double* lyi;         // [m] imag
  const double* lrating;     // [m]
  const double* lrating_inv; // [m] RN(1 / rating), 0 where the line has no rating (gs_div_by)
  // sparse block LU schedule (general / meshed networks)
  int32_t lu_n_piv, lu_n_slots, lu_n_orig, pad1;
  const int32_t* lu_piv_bus;     // [lu_n_piv] pivot bus, elimination order
  const int32_t* lu_nb_ptr;      // [lu_n_piv+1] neighbours of each pivot at elimination time
  const int32_t* lu_nb_bus;      // neighbour bus
  const int32_t* lu_nb_kj;       // slot of block (pivot, nb)
  const int32_t* lu_nb_jk;       // slot of block (nb, pivot)
  const int32_t* lu_pair_ptr;    // [lu_n_piv+1] ordered pairs (i, j) of neighbours, i != j or i == j
  const int32_t* lu_pair_ik;     // slot of (i, pivot)
  const int32_t* lu_pair_kj;     // slot of (pivot, j)
  const int32_t* lu_pair_ij;     // target slot: off-diagonal slot, or -(1+bus) for the diagonal of bus
  const int32_t* lu_orig_slot;   // [lu_n_orig] off-diagonal slots present in Y ...
  const int32_t* lu_orig_i;      // ... their (i, j) ...
  const int32_t* lu_orig_j;
  const int32_t* lu_orig_pos;    // ... and CSR position of (i, j)
  // the same elimination as a LEVEL schedule (pivots of a level commute): per wave and level,
  //   phase A  items (pivot bus, slot of (i, pivot) or -1): A_ik <- A_ik D_k^-1 in place (slot -1: only the singularity test of D_k)
  //   phase B  a stream of target records [target, count, count x (slot of (i, k), slot of (k, j) | pivot bus k)]: the wave owns the
  //            target for the level and subtracts all of the level's updates from it in registers -- target >= 0 an off-diagonal
  //            slot, -(1 + i) the diagonal block of bus i, -(1 + n + i) the right-hand side of bus i (second entry = pivot BUS)
  //   phase C  (back substitution, levels in descending order) pivot indices t into lu_piv_bus / lu_nb_*
  int32_t lu_n_levels, pad4;
  const int32_t* lu_a_ptr;       // [W * (lu_n_levels + 1)]
  const int32_t* lu_a;           // [.][2]
  const int32_t* lu_b_ptr;       // [W * (lu_n_levels + 1)] offsets into lu_b
  const int32_t* lu_b;
  const int32_t* lu_c_ptr;       // [W * (lu_n_levels + 1)]
  const int32_t* lu_c;
  // Iteration 0 of every Newton-Raphson solve starts from the flat start, where the Jacobian does not depend on the instance:
  // the handle factors it once (gs_create) and keeps the blocks here -- [lu_n_slots] off-diagonal blocks as the factorisation
  // leaves them (column blocks scaled by D_k^-1), then [n] diagonal blocks D_i, then one flag (a pivot was singular); read as
  // wave-uniform scalars.  Iteration 0 then only carries the right-hand side through: lu_r = phase B's right-hand-side records
  // alone, same format, per wave and level.  NULL: every solve factors for itself.
  const double* lu_flat;
  const int32_t* lu_r_ptr;       // [W * (lu_n_levels + 1)]
  const int32_t* lu_r;
  // dense partial-pivoting LU (reference-faithful linear solve, power_flow.py:187): unknown
  // order [theta(non-slack, ascending) ; Vm(pq, ascending)] exactly as power_flow.py:232-240
  int32_t dn_N, pad2;
  const int32_t* dn_th_idx;      // [n] row/column of bus i's theta unknown / P equation, -1 if none
  const int32_t* dn_vm_idx;      // [n] row/column of bus i's Vm unknown / Q equation, -1 if none
  // per-bus device lists for the injection build (reference accumulation order, grid_env.py:683-720)
  const int32_t* bl_ptr;     // [n+1] loads at bus
  const int32_t* bl_idx;
  const int32_t* bg_ptr;     // [n+1] generators at bus
  const int32_t* bg_idx;
  const int32_t* bb_ptr;     // [n+1] batteries at bus
  const int32_t* bb_idx;
  const double* load_base;   // [n_loads]
  const double* load_q;      // [n_loads] base * tan(acos(pf)) (base.py:283)
  const int32_t* gen_kind;   // [n_gens]
  const double* gen_cap;
  const double* gen_p0;
  const double* gen_p1;
  const double* gen_p2;
  const double* bat_cap;     // [n_bats]
  const double* bat_rating;
  const double* bat_eff;
};

// Row offsets into the per-group slab (units: rows of 64 doubles).
// A row family whose element i sits at base + 2 i: two such families with bases (even, even + 1) interleave, so that
// their elements i share one 16-byte slot per lane (GS_ELEM).  `R.E + i` keeps reading as before.
struct GsFam2 {
  int32_t base;
#if defined(__HIPCC__)
  __host__ __device__
#endif
  int operator+(int i) const { return base + 2 * i; }
};

struct GsRows {
  int32_t total;
  // solver inputs / outputs.  Paired (interleaved) families: (P, Q), (VM, VA), (FLOW, ENVLOAD), (E, F), (PC, QC),
  // (R0, R1), (X0, X1), (JR, JI)
  GsFam2 P, Q;               // [n] specified injections
  GsFam2 VM, VA;             // [n] polar state (solution)
  GsFam2 FLOW;               // [m] line P flow (paired with ENVLOAD)
  int32_t LOAD;              // [m] |S|/rating
  int32_t LOSSES, MAXMIS, ITERS, CONV, STATUS;   // scalars (stored as doubles)
  // solver scratch
  GsFam2 E, F;               // [n] rectangular voltage
  GsFam2 PC, QC;             // [n] calculated injections
  GsFam2 R0, R1;             // [n] mismatch (rhs)
  GsFam2 X0, X1;             // [n] Newton step
  int32_t RVM;               // [n] 1 / Vm (kept by the LDS forest solve)
  int32_t SV;                // [2n] inv(D) r          (the [2n] / [4n] blocks start on even rows: a bus's entries pair up)
  int32_t QV;                // [2n] child -> parent rhs contribution
  int32_t TB;                // [4n] inv(D) U
  int32_t CB;                // [4n] child -> parent diagonal contribution
  GsFam2 JR, JI;             // [n] FBS branch currents
  int32_t LU;                // [4 * lu_n_slots] off-diagonal blocks
  int32_t LUD;               // [4n] diagonal blocks
  int32_t DA;                // [N*N] dense Jacobian (dense kernel only)
  int32_t DB, DX, DPERM;     // [N] rhs, solution, row permutation
  // env state
  int32_t TIME, STEP, VIOL, TOTLOSS, EPREW, FREQ, IRR, WIND, TEMP, CLOUD, SEEDLO, SEEDHI;
  int32_t SOC, BATP;         // [n_bats]
  int32_t CURT, GENP;        // [n_gens] curtailment factor, uncurtailed renewable power
  GsFam2 ENVLOAD;            // [m] |flow|/rating (base.py:261-264), paired with FLOW
  // env outputs
  int32_t REWARD, TERM, TRUNC, VMAX, VMIN, VFLAGS /* [4] */;
  int32_t ACT;               // [action_dim] unpacked actions
  int32_t LOADP;             // [n_loads] realised load power of this step
};

// gs_k_nr_dense_mfma (kernels_dense.hip): Newton-Raphson with a dense block LU on the matrix cores, one workgroup per instance
// an entry of the Jacobian by 64 x 64 block (gs_k_nr_dense_mfma2): everything a thread needs to form one 2 x 2 block, 32 bytes
struct GsDenseEntry {
  int32_t ib, jb;            // row bus, column bus (equal: the diagonal block of the bus)
  int32_t dst;               // bits 0-15: (row in the block) * 66 + column in the block; 16 th_free(ib), 17 vm_free(ib), 18 th_free(jb), 19 vm_free(jb)
  int32_t pad;
  double g, b;               // Ybus entry (ib, jb); the diagonal entry for ib == jb
};
struct GsDenseArgs {
  int32_t n, na, NB, max_it;                 // buses, active (non-slack) buses, 64-wide panels (NP = 64 NB >= 2 na), iteration cap
  int32_t jacobian_exact, rows_total, pad0, pad1;
  double tol, alpha;
  const int32_t* act_bus;                    // [na] bus of active index a
  const int32_t* act_of;                     // [n] active index of a bus, -1 for the slack
  const int32_t* row_ptr; const int32_t* col; const double* G; const double* Bv; const double* Gd; const double* Bd;   // Ybus rows (GsTables)
  const int32_t* th_free; const int32_t* vm_free; const int32_t* fixed_v; const double* v_set;
  const int32_t* ent_ptr;                    // [NB + 1] Jacobian blocks whose COLUMN bus lies in panel p ...
  const int32_t* ent;                        // ... as (row bus i, column bus j, Ybus position of (i, j)) triples
  const int32_t* bent_ptr;                   // [NB * NB + 1] the same entries by block: (panel j, block row i) at j * NB + i (gs_k_nr_dense_mfma2) ...
  const GsDenseEntry* bent;                  // ... as records
  double* scratch;                           // [grid][(NB (NB - 1) / 2 L blocks + NB (NB - 1) / 2 U blocks + NB inverses) x 64 x 64]
  // The first Newton iteration starts from the flat start, where the Jacobian is the same for every instance: its block factors
  // are computed ONCE per handle (mode 1: one workgroup runs the factorisation below into `flat`) and iteration 0 of every solve
  // only substitutes with them -- a third of the factorisations of a typical three-iteration solve.
  double* flat;                              // [NB * NB blocks + 1 flag] or NULL
  // Iteration 0 as ONE matrix-vector product (block-row form): the inverse of the flat-start Jacobian, transposed ([column][row], NP x NP,
  // NP = 64 NB), inverted once on the host -- x = J0^-1 rhs instead of a forward and a back substitution over the flat-start factors
  // (seven block steps of two barriers and a round trip to the table each).  NULL: the factors.
  const double* jinv_t;
  int32_t mode, pad2;                        // 0 solve, 1 factor the flat-start Jacobian into `flat`
  unsigned long long* stamps;                // diagnostic (gs_debug_stamps): cycles per phase of workgroup 0, NULL = off
  GsRows R;
};


// gs_k_nr_sparse_lds (kernels_sparse.hip): Newton-Raphson with the sparse 2x2-block LU of an instance in LDS, one wavefront per
// instance, several wavefronts (instances) per workgroup sharing ONE copy of the schedule and of the Ybus rows in LDS.  The
// schedule is the level schedule of linsolve_lu (kernels_solve.hip) without its split over waves: per level a list of phase-A
// items and a list of phase-B records, each taken by a lane.
struct GsSparseArgs {
  int32_t n, n_slots, n_orig, n_piv, n_levels, max_it, jacobian_exact, rows_total;
  double tol, alpha;
  // staged into LDS once per workgroup: `ipack` (int32) and `dpack` (double), the arrays below at these offsets
  const int32_t* ipack; const double* dpack;
  int32_t ipack_n, dpack_n;
  int32_t o_row_ptr, o_col, o_th_free, o_vm_free, o_fixed_v;        // Ybus rows (CSR), unknown masks
  int32_t o_piv_bus;                                                // [n_piv] bus of pivot t
  int32_t o_nb_ptr, o_nb_bus, o_nb_kj;                              // per pivot: the neighbours left when it is eliminated, slot of A_kj
  int32_t o_a_ptr, o_a;                                             // [n_levels + 1]; (pivot bus k, slot of A_ik or -1: the pivot's singularity test alone)
  int32_t o_b_ptr, o_b_rec, o_b_pair;                               // [n_levels + 1] record ranges; (target, count, first pair); (slot of L_ik, slot of A_kj | bus k)
                                                                    // target: >= 0 an off-diagonal slot, -1 - i the diagonal block of bus i, -1 - n - i the right-hand side of bus i
  int32_t o_r_ptr, o_r_rec, o_r_pair;                               // the right-hand-side records alone (iteration 0 with the flat-start factors)
  int32_t o_c_ptr, o_c;                                             // [n_levels + 1]; pivots (index t) of a level, for the back substitution
  int32_t od_G, od_B, od_Gd, od_Bd, od_vset;                        // offsets into dpack
  int32_t waves, wave_bytes;                                        // wavefronts (instances) per workgroup; LDS bytes of one wavefront's instance
  const int32_t* orig_slot; const int32_t* orig_i; const int32_t* orig_j; const int32_t* orig_pos;   // [n_orig] the network's own off-diagonal blocks (global: read once per assembly, no chains)
  const double* flat;                                  // [4 (n_slots + n) + 1] factors of the flat-start Jacobian + singular flag, or NULL
  double* flat_out;                                    // mode 1: where one wavefront leaves them
  int32_t mode, pad0;
  unsigned long long* stamps;
  GsRows R;
};

struct GsPackArgs {
  const int32_t* map;      // obs column -> slab row, or -(1 + constant index)
  const double* cst;
  double* out;             // [B][obs_dim]
  int32_t obs_dim, tiles_per_pass;   // 64-column tiles staged in LDS per pass
  int32_t early_pass0, pair_ok;      // early_pass0: no column of the first pass is written by the epilogue's scalar part;
                                     // pair_ok: obs_dim and the block of constants are even (two columns per lane)
  int32_t skip0, skip1;              // columns [skip0, skip1) are per-instance constants (the static load powers of
                                     // grid_env.py:769-770): written at reset, left alone by the step (skip0 == skip1: none)
  int32_t lean, pad0;                // second-generation kernels: the (|V|, angle) and (flow, |P| / rating) row pairs are NOT written --
                                     // they are columns of the observation block the step writes anyway (a third of its bytes twice);
                                     // the host restores the rows from the block when somebody asks for them (gridstep_abi.hip, ensure_rows)
};

// ---- "flow2" kernels (kernels_flow2.hip): 32 instances per workgroup, the two halves of a wavefront on DIFFERENT buses ----
// A 64-instance group fills a compute unit's LDS with its per-bus slots, so a batch of 8192 instances used 128 of the
// 256 CUs.  Here a workgroup owns HALF a slab group (instances g * 64 + hs * 32 + l, l = lane & 31) and the lanes
// 32..63 of every wavefront work on a second bus (line, load quad ...) of the same 32 instances: every bus-parallel
// phase issues half the instructions per workgroup, and there are twice as many workgroups.  What used to be
// wave-uniform (bus index, impedance, child list, row index) is now uniform per HALF: it lives in vector registers,
// loaded from the per-(wave, item, half) records below.
#define GS_F2_PITCH 33            /* 16-byte entries per LDS slot: 32 lanes + 1 (transposed reads conflict-free) */
#define GS_F2_ITEMS 4             /* buses per half wave: 16 waves x 2 halves x 4 = 128 positions of the forest's preorder */
#define GS_F2_WAVES 16
#define GS_F2N_WAVES 8            /* the Newton-Raphson member of the family: 8 waves x 2 halves x 8 buses (its bus state needs the registers) */
#define GS_F2N_ITEMS 8
#define GS_F2S_IW 8               /* small feeders: 8 instances per workgroup, the eight sub-groups of a wavefront on eight buses */
#define GS_F2H_IW 16       // half-size member: 16 instances per workgroup, two workgroups per CU
#define GS_F2H_WAVES 8
#define GS_F2H_ITEMS 4
#define GS_F2X_WAVES 8        // wide member: 16 instances per workgroup, eight buses per sub-group (up to 256 buses)
#define GS_F2X_ITEMS 8
#define GS_F2S_WAVES 2            /* sweeps: 2 waves x 8 sub-groups x 1 bus = 16 positions */
#define GS_F2S_ITEMS 1
#define GS_F2NS_WAVES 4           /* Newton-Raphson: 4 waves x 1 item, each a group of 8 buses of one level (2 x 2: 124 M env-steps/s on config 2; 4 x 1: 147 M -- the load draws get waves of their own) */
#define GS_F2NS_ITEMS 1
#define GS_F2M_WAVES 4            /* the meshed Newton-Raphson member: 4 waves x up to 9 rows of 8 sub-groups */
#define GS_F2M_ITEMS 10
#define GS_F2_CHILDREN 8          /* children per bus in the Newton-Raphson kernel's LDS child tables */
struct GsF2Rec {                  // one preorder position p = ((wave * 2 + half) * GS_F2_ITEMS + item); 96 bytes
  int32_t bus, parent, flags, last;         // slot indices; flags: bit0 active, bit1 root (parent is the slack bus); last: the bus at the LAST position of this bus's subtree
  int32_t nl, l0, l1, ng;                   // devices at the bus, reference accumulation order (grid_env.py:689-718)
  int32_t g0, g1, nb, b0;
  int32_t b1, level, pad1, pad2;
  double zr, zi, yr, yi;                    // branch to the parent: z = 1 / y
};
// ---- the meshed Newton-Raphson member (F2_NRM, mesh_schedule.h): one (wavefront, row, sub-group) of the block elimination ----
#define GS_MESH_ACC 4             /* most accumulators per target = entries per pull list */
struct GsMeshItem {                     // 192 bytes
  int32_t vk_off, vj_off;               // LDS byte offsets (without the lane's share): voltage slot of the pivot bus / of neighbour j_t
  int32_t xk_off, xj_off;               // back substitution: where lane 0 leaves x_k / where lane t finds x_jt
  int32_t flags;                        // GS_MESH_F_*
  int32_t cq_off;                       // CQ(k -> j_t) accumulator
  int32_t adj_ptr, bus;                 // first entry of the pivot bus's Ybus row (lane 0); the pivot bus (-1: idle)
  double ykj_g, ykj_b, ykk_g, ykk_b;    // Y(k, j_t) (0 where the pair is fill); Y(k, k) (lane 0; (0, -1) elsewhere: an identity-like diagonal block)
  int32_t mout[8];                      // accumulator of M(k -> (j_t, j_t')) for t' = 0 .. g - 1 (entry t: the C part of cq_off)
  int32_t cq_in[GS_MESH_ACC], rw_in[GS_MESH_ACC], cl_in[GS_MESH_ACC];      // pull lists (padded with the ZERO message): into (D_k, r_k), A(k, j_t), A(j_t, k)
  int32_t nbr, pos;                     // neighbour bus (-1: none); index of the item
  int32_t pad[10];                      // pad[0]: the group's slot in the (D^-1, s) exchange (GS_MESH_W_GSLOT)
};
enum {
  GS_MESH_F_PIVOT = 1,          // lane 0 of a group that eliminates a bus
  GS_MESH_F_NBR = 2,            // the lane has a neighbour (blocks, T, messages)
  GS_MESH_F_SLACKPOS = 4,       // takes part in the mismatch pass only: the slack's share of the losses sum
  GS_MESH_F_HV0_SHIFT = 4,      // bits 4-7: first sub-group of the lane's group
  GS_MESH_F_T_SHIFT = 8,        // bits 8-11: t
  GS_MESH_F_G_SHIFT = 12,       // bits 12-15: group size (1, 2, 4, 8)
  GS_MESH_F_RMW_SHIFT = 16      // bits 16-23: output t' adds to what its accumulator holds (else: first producer, plain write)
};

// The same item as the kernel reads it: 16 words.  Offsets are slot / unit numbers (a voltage slot is (IW + 1) * 16 bytes, a unit of
// the message region 16 * IW bytes; unit 0 = the ZERO message, 3 = DUMMY, 6 + i = the body's unit i; x slot of bus b = unit 6 + b).
enum {
  // words 0-3: what a row needs BEFORE its level's messages (read a row ahead)
  GS_MESH_W_BUS_NBR = 0,        // pivot bus slot | neighbour slot << 16 (idle: the ONE slot)
  GS_MESH_W_FLAGS = 1,          // GS_MESH_F_* in bits 0-23; bits 24-31: neighbour entries of the pivot bus (lane 0 of a pivot / the slack's position; 0 elsewhere)
  GS_MESH_W_PAIR_CQ = 2,        // Ybus pair of (bus, neighbour) (n_pairs: none) | unit of CQ(k -> j_t) << 16
  GS_MESH_W_DIAG_ADJ = 3,       // slot whose diagonal Ybus entry the lane uses | first neighbour entry of the pivot bus << 16
  // words 4-15: what it needs after them (read at the row's start)
  GS_MESH_W_CQIN = 4,           // 2 words cq_in, then 2 words rw_in, 2 words cl_in: units, two per word
  GS_MESH_W_MOUT = 10,          // 4 words: units of M(k -> (j_t, j_t')), two per word
  GS_MESH_W_GSLOT = 14,         // which of the row's groups of two or more lanes this lane's is (0 .. 3): its slot in the (D^-1, s) exchange
  GS_MESH_WORDS = 16
};
#define GS_MESH_F_NADJ_SHIFT 24

struct GsF2Tables {
  const GsF2Rec* recs;            // [GS_F2_WAVES * 2 * GS_F2_ITEMS]
  const int32_t* anc;             // sweeps: [n_jump][n_slots] 2^r-th ancestor of every slot on its way to the slack, ZERO beyond;
                                  // Newton-Raphson: child buses [n][8], child ring slots [n][8], child counts [n_slots], then at
                                  // pos_off (a multiple of 4) per position (bus, parent, own ring slot, parent's ring slot); n_anc_ints in all
  const double* zbus;             // sweeps: [n_slots][2] impedance of the branch from each bus to its parent (0 where there is none);
                                  // Newton-Raphson: [n_slots][4] (G_ip, B_ip, G_ii, B_ii)
  int32_t n_slots;                // n + 3: buses, then ZERO (0, 0), ONE (1, 0), DUMMY
  int32_t slack;                  // slot of the slack bus
  int32_t n_jump;                 // rounds of the forward sweep's pointer jumping: ceil(log2(depth)), rounded up to even
  int32_t n_levels, pos_off, n_anc_ints, wg_offset;   // Newton-Raphson: levels of the tree below the slack; layout of `anc`
  int32_t off_tile, off_anc, off_z, off_env, off_red, off_atom, lds_bytes;     // LDS byte offsets (slots at 0)
  int32_t off_prof, ring_zero;    // ring_zero (Newton-Raphson): index of the ring entry that stays zero;     // the 24 hourly factors of the load profile (dynamics.py:37-44), copied from kDailyProfile at kernel start
  int32_t env_genp, env_curt, env_batp, env_soc;      // row indices inside the env area ([row][32 lanes] doubles)
  // buses with a voltage set point (normally the slack alone): slot and |V|; the first inline, the rest through the arrays
  int32_t n_fixed, fixed_slot0;
  double fixed_val0;
  const int32_t* fixed_slot; const double* fixed_val;
  // Newton-Raphson: iteration 0 starts from the flat start, where the Jacobian -- and with it every D_i^-1, T_i and the
  // branch's lower block L_i of the elimination -- does not depend on the instance.  nrflat: [positions][16] doubles per item
  // position (P calc, Q calc, its share of the losses sum, -, D^-1 (4), T (4), L (4)), written ONCE per handle by the kernel
  // itself (nrflat_mode 1: one workgroup, gs_create) and read by iteration 0 of every later step (mode 2), which then only
  // carries its right-hand side up and down the tree.  Mode 0 / NULL: every iteration eliminates for itself.
  double* nrflat; int32_t nrflat_mode, pad_nrflat;
  // the meshed member: items [NW * NI * 8], rowinfo [NW * NI][4] (level | -1, g | ncq << 8 | nrw << 16 | ncl << 24, nadj, -), the
  // Ybus rows of the pivot buses (voltage slot offset; (G, B)), and its per-wave exchange scratch in LDS
  const int32_t* mesh_items; const int32_t* mesh_rowinfo;      // packed items (GS_MESH_W_*), [NW * NI * 8][16]
  int32_t off_scr, mesh_nz;                                    // exchange scratch; doubles of the Ybus tables staged at off_z (pairs, then diagonal per slot)
  int32_t mesh_pairs, mesh_off_p;                              // P_spec by voltage slot, [slot][8 instances] doubles
  // Iteration 0 of the meshed member as a matrix product (NULL: by elimination with the flat-start table).  At the flat start the
  // Jacobian does not depend on the instance, so the first Newton step is a constant linear map of the instance's injections:
  //   x = J0^-1 (S_spec - S_calc(flat)) = W [P_spec; 1]     (Q_spec = 0 at every PQ bus),
  // W = [the angle-equation columns of J0^-1 | the constant term], (2 (n - 1)) x n, inverted ONCE on the host (gs_create) and stored in
  // the operand order of v_mfma_f64_16x16x4: [16 row tiles][mesh_w_steps k-steps][64 lanes], A[row = lane & 15][k = lane >> 4].
  const double* mesh_w; int32_t mesh_w_steps, mesh_slack;
};

// gs_k_rollout_post (kernels_env.hip): bookkeeping after step t of gs_rollout
struct GsRolloutPostArgs {
  double* rew; uint8_t* done;            // [T][B]
  double* obs_next;                      // [B][obs_dim], slot t + 1 of the observation sequence
  const int32_t* map; const double* cst; // observation column -> slab row, or -(1 + constant index)
  int32_t* term_count; int32_t* term_idx; double* term_obs; int32_t term_cap;
  int32_t obs_dim, t, B;
};

// The rollout collector's bookkeeping when it is fused into the step kernel (kernels_flow2.hip): at the START of step t the
// kernel files the instances that step t - 1 finished (terminal observation to the side list, in-place reset, fresh
// observation into slot t), at its end it writes reward and done flags of step t.  active == 0: an ordinary step.
struct GsRolloutStep {
  double* rew; uint8_t* done;            // [T][B]
  double* obs_prev;                      // [B][obs_dim] slot t of the observation sequence (what this step starts from)
  const int32_t* map; const double* cst; // observation column -> slab row, or -(1 + constant index)
  int32_t* term_count; int32_t* term_idx; double* term_obs;
  int32_t term_cap, obs_dim, t, active;
};

#define GS_STAMP_BLOCKS 2048
struct GsSolveCfg {
  double tolerance, alpha;
  int32_t max_iterations, jacobian_exact;
  unsigned long long* stamps;   // diagnostic: per-phase cycle sums of block 0 / wave stamp_wave (NULL = off); behind the 16
                                // sums, when block_times is set: (start, end) of every workgroup on the 100 MHz real-time clock
  int32_t stamp_wave, block_times;
};

struct GsEnvCfg {
  double timestep, v_min, v_max, f_min, f_max, safety_penalty, H, D, f0, power_base, inv_power_base;
  int32_t episode_length, stochastic_loads, weather_variation, fbs_warm_start;
  int64_t first_instance;
};

// ---- post-step checks (kernels_checks.hip) ----
struct GsChecksCfg {
  double c_vlo, c_vhi, c_flo, c_fhi, c_load, c_rocv, c_rocf, dt;                       // SafetyChecker
  double m_vlo, m_vhi, m_flo, m_fhi, m_load, m_evlo, m_evhi, m_eflo, m_efhi;           // SafetyMonitor
  double q_tol;                                                                          // quality gate
  int32_t n, m, rows_total;
  int32_t row_vm, row_cload, row_qload, row_flow, row_freq, row_conv, row_iters, row_maxmis;   // row_vm, row_flow: stride 2
  int32_t stride_cload;
};

// The same checks evaluated inside the step kernel's epilogue (gs_checks_set_fused): its bus and line loops already
// hold every value, so the safety checks of a step cost a few comparisons instead of a launch that re-reads the state.
struct GsFusedChecks {
  GsChecksCfg C;             // thresholds; stride_cload == 2: the environment's |P| / rating is what the limits apply to
  double* prev; int32_t* state; int32_t* out_i; double* out_f; uint8_t* bus_mask; uint8_t* line_mask;
  int32_t enabled, Bp;
};

