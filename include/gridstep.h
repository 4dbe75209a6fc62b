/* gridstep.h -- C ABI of libgridstep.so: MI355X-native batched AC power flow + env.step().
 *
 * This is the drop-in boundary for the hot path of danieleschmidt/grid-fed-rl-gym.  The
 * reference has no FFI layer: the path sits behind two duck-typed Python plug points, and
 * each entry point below names the reference interface it replaces (file:line relative to
 * /root/reference/grid_fed_rl/).  The Python binding a maintainer would add is a ctypes
 * stub; grid_fed_rl_gym_amd/_lib.py is that stub.
 *
 * Conventions
 *  - plain pointers and sizes only; every array is caller-owned, dense, C order, float64
 *    unless stated; batch axis first:  P_spec[B][n], obs[B][obs_dim], actions[B][action_dim].
 *  - every function returns 0 (GS_OK) or a negative GS_E* code; gs_last_error() gives text.
 *  - per-INSTANCE numerical failure never fails the call: it sets status[b]
 *    (0 converged, 1 iteration cap, 2 singular Jacobian, 3 non-finite mismatch), mirroring
 *    the reference's "return converged=False" convention (environments/power_flow.py:186-190).
 *  - one handle = one device = one HIP stream.  A handle is not thread-safe; distinct
 *    handles are independent.
 *  - there is no CPU fallback: gs_create fails with GS_E_NO_DEVICE when no GPU is present.
 */
#ifndef GRIDSTEP_H
#define GRIDSTEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 1

enum {
  GS_OK = 0,
  GS_E_INVALID = -1,     /* bad argument / shape / enum value                       */
  GS_E_NO_DEVICE = -2,   /* no HIP device, or device index out of range             */
  GS_E_HIP = -3,         /* a HIP runtime call failed                               */
  GS_E_TOPOLOGY = -4,    /* network unusable for the requested solver (e.g. FBS on a mesh) */
  GS_E_STATE = -5,       /* call sequence error (e.g. step before reset)            */
  GS_E_COMM = -6,        /* RCCL not loadable / communicator failure                */
  GS_E_NOMEM = -7
};

enum { GS_BUS_PQ = 0, GS_BUS_PV = 1, GS_BUS_SLACK = 2 };          /* base.py:210 bus_type   */
enum { GS_JACOBIAN_AS_CODED = 0, GS_JACOBIAN_EXACT = 1 };         /* power_flow.py:247-248  */
enum { GS_ZERO_Z_OPEN = 0, GS_ZERO_Z_EPSILON = 1 };               /* power_flow.py:63       */
enum { GS_SOLVER_NR = 0, GS_SOLVER_FBS = 1 };
/* AUTO: exact Jacobian -> TREE on radial networks, SPARSE_LU on meshed ones (2x2-block pivots,
 * no row exchanges: safe because the exact diagonal blocks are rotation-like);
 * as-coded Jacobian -> DENSE_PIVOT (partial pivoting like LAPACK dgesv, power_flow.py:187),
 * because the as-coded diagonal blocks can be exactly singular (e.g. a leaf fed through r = x). */
enum { GS_LINSOLVE_AUTO = 0, GS_LINSOLVE_TREE = 1, GS_LINSOLVE_SPARSE_LU = 2, GS_LINSOLVE_DENSE_PIVOT = 3,
       /* dense block LU on the matrix cores, one workgroup per instance (exact Jacobian, at most 128 non-slack buses);
        * what AUTO takes for a meshed network whose sparse LU would fill in (more than a quarter of all blocks) */
       GS_LINSOLVE_DENSE_MFMA = 4,
       /* sparse 2x2-block LU with all blocks of an instance in LDS, one wavefront per instance (meshed networks of at most 256
        * buses whose blocks fit 32 KB).  An alternative to SPARSE_LU that moves 1/20 of its bytes and is slower on the feeders
        * measured: never AUTO's choice */
       GS_LINSOLVE_SPARSE_LDS = 5 };
enum { GS_GEN_SOLAR = 0, GS_GEN_WIND = 1 };
enum { GS_STATUS_OK = 0, GS_STATUS_MAX_ITER = 1, GS_STATUS_SINGULAR = 2, GS_STATUS_NAN = 3,
       GS_STATUS_FALLBACK_LINEAR = 4 /* answer replaced by gs_fallback_linear; converged = 1 as the reference's linear solver reports */ };

/* Network + devices, flattened.  Replaces the Bus/Line/Load object lists the reference
 * passes to solve() (environments/base.py:197-295, power_flow.py:38-46) and the feeder
 * containers (feeders/base.py:44-47).  Bus references are 0-based list positions (the
 * reference's bus_map, power_flow.py:54). */
typedef struct gs_topology {
  int32_t struct_size;            /* = sizeof(gs_topology) */
  int32_t n, m;
  const int32_t* from_bus;        /* [m] */
  const int32_t* to_bus;          /* [m] */
  const double* r;                /* [m] per unit */
  const double* x;                /* [m] per unit */
  const double* rating;           /* [m] VA, as the reference stores it */
  const uint8_t* bus_type;        /* [n] GS_BUS_* */
  const double* v_set;            /* [n] magnitude set-point of slack / pv buses */
  int32_t n_loads;
  const int32_t* load_bus;        /* [n_loads] */
  const double* load_base;        /* [n_loads] base_power (base.py:280) */
  const double* load_pf;          /* [n_loads] power factor (base.py:281) */
  int32_t n_gens;
  const int32_t* gen_bus;         /* [n_gens] */
  const int32_t* gen_kind;        /* [n_gens] GS_GEN_* */
  const double* gen_cap;          /* [n_gens] capacity */
  const double* gen_p0;           /* solar: efficiency   | wind: cut-in speed  (dynamics.py:116,150) */
  const double* gen_p1;           /* solar: panel area   | wind: rated speed   */
  const double* gen_p2;           /* solar: unused       | wind: cut-out speed */
  int32_t n_bats;
  const int32_t* bat_bus;         /* [n_bats] injection bus (reference hard-codes id 2, grid_env.py:714) */
  const double* bat_cap;          /* [n_bats] dynamics.py:178 */
  const double* bat_rating;       /* [n_bats] dynamics.py:179 */
  const double* bat_eff;          /* [n_bats] dynamics.py:180 */
} gs_topology;

/* Solver + environment configuration.  Replaces the constructor kwargs of
 * NewtonRaphsonSolver (power_flow.py:79-87) and GridEnvironment (grid_env.py:161-174) and
 * GridDynamics (dynamics.py:233-238). */
typedef struct gs_config {
  int32_t struct_size;            /* = sizeof(gs_config) */
  int32_t solver_kind;            /* GS_SOLVER_* */
  int32_t jacobian_mode;          /* GS_JACOBIAN_* */
  int32_t zero_z_mode;            /* GS_ZERO_Z_* */
  int32_t linear_solver;          /* GS_LINSOLVE_* */
  int32_t max_iterations;         /* power_flow.py:82 (default 50) */
  int32_t episode_length;         /* grid_env.py:165 */
  int32_t stochastic_loads;       /* grid_env.py:166 (Philox stream) */
  int32_t weather_variation;      /* grid_env.py:168 */
  int32_t waves_per_group;        /* 0 = auto; 1,2,4,8,16: waves cooperating on one 64-instance group */
  int32_t fbs_warm_start;         /* 0 (default): every step solves from the flat start, as the reference's solve() does
                                     (power_flow.py:125-134); 1: the sweep solver starts from the previous step's voltages --
                                     same tolerance, fewer sweeps; an option, never the measured headline */
  double tolerance;               /* power_flow.py:81 (default 1e-6).  Sweep solver: the summed mismatch is held against
                                     tolerance / 2; below 1e-10 a handle runs the first-generation sweep kernels (double-precision
                                     comparison; the second generation keeps the sum in 2^-44 pu fixed point), gs_describe says so */
  double acceleration_factor;     /* power_flow.py:83 (default 1.0) */
  double timestep;                /* grid_env.py:164 */
  double v_min, v_max;            /* grid_env.py:170 */
  double f_min, f_max;            /* grid_env.py:171 */
  double safety_penalty;          /* grid_env.py:172 */
  double inertia_H, damping_D, f_nominal;   /* dynamics.py:235-237 */
  double power_base;              /* injections are divided by this before the solve; 1.0 = as coded (F4) */
} gs_config;

/* Host destination pointers for one batched solution; any pointer may be NULL (skipped).
 * Field-for-field the reference's PowerFlowSolution (power_flow.py:12-22) with a batch axis. */
typedef struct gs_solution_view {
  double* bus_voltages;           /* [B][n] */
  double* bus_angles;             /* [B][n] rad, wrapped to (-pi, pi] like np.angle */
  double* line_flows;             /* [B][m] P from->to */
  double* line_loadings;          /* [B][m] |S|/rating */
  double* losses;                 /* [B] */
  double* max_mismatch;           /* [B] */
  int32_t* iterations;            /* [B] last loop index + 1 (power_flow.py:204) */
  uint8_t* converged;             /* [B] */
  int32_t* status;                /* [B] GS_STATUS_* */
} gs_solution_view;

/* Host destination pointers for the per-step info dict (grid_env.py:610-617); NULLs skipped. */
typedef struct gs_info_view {
  uint8_t* power_flow_converged;  /* [B] */
  double* max_voltage;            /* [B] */
  double* min_voltage;            /* [B] */
  double* total_losses;           /* [B] solution.losses of this step */
  uint8_t* violations;            /* [B][4] voltage_high, voltage_low, frequency_high, frequency_low (base.py:153-167) */
  int32_t* constraint_violations; /* [B] running count (grid_env.py:586) */
  int32_t* current_step;          /* [B] */
  double* episode_reward;         /* [B] */
  int32_t* iterations;            /* [B] */
  int32_t* status;                /* [B] */
} gs_info_view;

typedef struct gs_handle gs_handle;

/* ---- library ------------------------------------------------------------------------- */
int gs_version(void);
/* 1 if the library was built with `make EXPERIMENTS=1`: the kernel members and environment switches that exist to measure
 * alternatives (linear_solver sparse_lds, the 32-instance sweep member, table-layout and launch-shape switches) are present.
 * The default build carries what AUTO can reach and the switches the parity tests compare handles with (csrc/gs_internal.h). */
int gs_build_experiments(void);
int gs_device_count(void);
/* text of the last error on this thread (handle may be NULL for creation errors) */
const char* gs_last_error(const gs_handle* h);

/* ---- lifetime ------------------------------------------------------------------------ */
/* Compiles the topology (Ybus CSR, elimination schedule), allocates all device memory for
 * `batch` instances on `device`.  Replaces solver/env construction (power_flow.py:79,
 * grid_env.py:161-241).  `first_instance` is the global index of this handle's instance 0
 * (rank * B_local when a batch is sharded over GPUs); it only feeds the RNG counters. */
int gs_create(const gs_topology* topo, const gs_config* cfg, int32_t batch, int32_t device,
              int64_t first_instance, gs_handle** out);
void gs_destroy(gs_handle* h);
int gs_dims(const gs_handle* h, int32_t* n, int32_t* m, int32_t* obs_dim, int32_t* action_dim,
            int32_t* state_dim, int32_t* batch);
/* how the topology was compiled: linear solver chosen, tree depth, fill, waves per group */
int gs_describe(const gs_handle* h, char* buf, int32_t buflen);
int gs_synchronize(gs_handle* h);

/* ---- solver plug point: NewtonRaphsonSolver.solve (power_flow.py:89-211) ---------------
 * P_spec[B][n] is the net specified injection (generation - load, what :112-121 builds from
 * the two dicts); Q_spec may be NULL (the reference never injects reactive power, :107). */
int gs_solve(gs_handle* h, const double* P_spec, const double* Q_spec, const gs_solution_view* out);
/* device-resident variant for measurement: upload once, solve many times, download once */
int gs_upload_injections(gs_handle* h, const double* P_spec, const double* Q_spec);
int gs_solve_device(gs_handle* h);
int gs_download_solution(gs_handle* h, const gs_solution_view* out);

/* ---- fallback for rejected load flows: LinearApproximationSolver.solve (robust_power_flow.py:336-398), the
 * "linear_approximation" stage of AdvancedRobustPowerFlowSolver.solve (:523-613) ------------------------------
 * Overwrites the solution rows (voltages, angles, flows, loadings, losses, converged = 1, iterations = 1,
 * max_mismatch = 0, status = GS_STATUS_FALLBACK_LINEAR) of the instances selected by mask[B] (!= 0), or, with
 * mask == NULL, of the instances whose last solve / step did not converge; every other instance keeps its rows.
 * load_w / gen_w: [B][n] totals of the reference's `loads` / `generation` dicts per bus (0 = no entry), in the
 * reference's units (W); both NULL = take them from the environment state on the device (load powers, curtailed
 * renewables, battery powers: grid_env.py:683-720).  total_load / total_gen: [B] sums of the dict values in the
 * caller's dict order, or NULL = summed over the buses in index order (host arrays) / in the order the
 * reference's dicts are filled (device state).  applied_out: NULL or [B], 1 where the answer was replaced;
 * n_applied: NULL or their count.  Follow with gs_download_solution / gs_checks_run / gs_download_step as usual. */
int gs_fallback_linear(gs_handle* h, const double* load_w, const double* gen_w, const double* total_load,
                       const double* total_gen, const uint8_t* mask, uint8_t* applied_out, int32_t* n_applied);

/* ---- env plug point: GridEnvironment.reset/step (grid_env.py:360-408, 410-619) and their
 * batched form VectorizedEnvironment.reset/step (utils/parallel_environment.py:309-355) --- */
/* seeds: [B] per-instance RNG seeds, or NULL = the stream runs on, as the reference's reset(seed=None) leaves its
 * global generators running (grid_env.py:366-369): every reset instance gets the next seed of its chain, one Philox
 * call keyed by the seed it holds (0 on a fresh handle) with counter (global instance, 0, 'RSED') -- so consecutive
 * episodes differ while gs_reset(seeds) stays exactly reproducible; mask: NULL (all) or [B] (reset where != 0);
 * obs_out: NULL or [B][obs_dim]. */
int gs_reset(gs_handle* h, const uint64_t* seeds, const uint8_t* mask, double* obs_out);
int gs_step(gs_handle* h, const double* actions, double* obs, double* reward,
            uint8_t* terminated, uint8_t* truncated, const gs_info_view* info);
/* Page-locked host memory for the arrays gs_step / gs_download_step / gs_reset / gs_rollout_download fill.  The reference's
 * step() returns fresh NumPy arrays every call (grid_env.py:563-619); at [8192][684] doubles that is 45 MB of first-touch
 * page faults plus a staged copy per step.  A caller that hands gs_step buffers from gs_host_alloc gets the copy at the
 * link's rate and asynchronously (measured: DESIGN.md section 5, `with_host_io`).  Plain host allocations, owned by the
 * caller: gs_host_free releases them; they outlive any handle. */
int gs_host_alloc(void** out, size_t bytes);
int gs_host_free(void* p);

/* device-resident variant: K action batches [K][B][action_dim] staged in HBM, stepped by index */
int gs_upload_actions(gs_handle* h, const double* actions, int32_t n_batches);
int gs_step_device(gs_handle* h, int32_t action_batch_index);
int gs_download_step(gs_handle* h, double* obs, double* reward, uint8_t* terminated,
                     uint8_t* truncated, const gs_info_view* info);
/* The same with the observation block as FLOAT32 [B][obs_dim] -- the dtype the reference declares for its observation space
 * (grid_env.py:346, `Box(..., dtype=np.float32)`; the values its step() returns are Python floats, which is what gs_step hands out).
 * Rounded to nearest on the device; every other output as in gs_step / gs_download_step.  Opt-in: half the bytes over the link. */
int gs_step_f32(gs_handle* h, const double* actions, float* obs, double* reward, uint8_t* terminated,
                uint8_t* truncated, const gs_info_view* info);
int gs_download_step_f32(gs_handle* h, float* obs, double* reward, uint8_t* terminated,
                         uint8_t* truncated, const gs_info_view* info);
/* Host observation arrays that are handed to gs_step / gs_download_step again and again (the recycled page-locked sets of the
 * Python environment): gs_host_obs_bind writes the constant columns of the observation -- the static load powers of
 * grid_env.py:769-770, a third of the row on the 123-bus feeder -- into `obs` once and remembers the address; later downloads
 * into that address move the changing columns only (one pitched copy: in memory order the changing columns of row r behind the
 * constants and those of row r + 1 in front of them are one run), the same bytes as a whole-row download would leave.
 * The caller must not modify the constant columns of a bound array (bind again if it did).  After gs_reset. */
int gs_host_obs_bind(gs_handle* h, double* obs);
int gs_host_obs_unbind(gs_handle* h, double* obs);

/* A consumer that lives on the same GPU (a policy network) steps the environment without any host copy: it reads the
 * observation block and the reward / flag arrays through device pointers and hands back a device pointer to its actions.
 * Streams are passed as `hipStream_t` cast to `void*`; NULL means "the caller synchronises itself"; the legacy default
 * stream (handle 0 -- PyTorch's current stream unless the caller created one) is passed as hipStreamLegacy, (void*)1.
 *   gs_step_device_ptr   one env step with actions[B][action_dim] (float64, C order) in DEVICE memory of the handle's GPU; when
 *                        `producer_stream` is given, the step waits (on the device) for the work queued on it so far.
 *   gs_step_device_view  device pointers to what the last step left: observations[B][obs_dim] (one of the handle's two
 *                        observation buffers: valid until the next-but-one step), reward[B], terminated[B], truncated[B]
 *                        (refreshed by this call; valid until the next call); when `consumer_stream` is given it is made to
 *                        wait (on the device) for the step, otherwise the call returns after the step has finished. */
typedef struct gs_step_device_out {
  double* observations;
  double* reward;
  uint8_t* terminated;
  uint8_t* truncated;
  int32_t B, obs_dim;
} gs_step_device_out;
int gs_step_device_ptr(gs_handle* h, const double* d_actions, void* producer_stream);
int gs_step_device_view(gs_handle* h, gs_step_device_out* out, void* consumer_stream);

/* ---- device-resident rollout collection: collect_random_data(env, num_steps) and the five arrays GridDataset is
 * built from (algorithms/base.py:268-298, 180-205) ------------------------------------------------------------
 * T env steps back to back on the device -- no host round trip between them: the step kernel of step t writes its
 * observation block straight into slot t + 1 of obs_seq[T + 1][B][obs_dim], a small kernel behind it files reward and
 * done flags and resets the instances that finished (terminated or truncated) in place, exactly where the reference
 * calls env.reset() (base.py:289-290); their seed is the next one of the instance's chain (see gs_reset).
 * policy: GS_POLICY_RANDOM -- uniform actions in (-1, 1) drawn on the device (the reference samples
 * env.action_space; here Philox keyed by policy_seed, counter (global instance, t, action / 4, 'ACTN'), word k of a
 * call = action 4 q + k = 2 (r + 1/2) 2^-32 - 1); GS_POLICY_UPLOADED -- actions[T][B][action_dim] from the caller.
 * The call is asynchronous; the environment afterwards stands where T calls of gs_step (+ resets) would have left it. */
enum { GS_POLICY_UPLOADED = 0, GS_POLICY_RANDOM = 1 };
int gs_rollout(gs_handle* h, int32_t T, int32_t policy, uint64_t policy_seed, const double* actions);
/* host copies in the reference's layout, transition index = t * B + b; any pointer may be NULL */
typedef struct gs_rollout_view {
  double* observations;           /* [T][B][obs_dim] what each step started from */
  double* actions;                /* [T][B][action_dim] */
  double* rewards;                /* [T][B] */
  double* next_observations;      /* [T][B][obs_dim] (the terminal observation where the transition ended an episode) */
  uint8_t* terminals;             /* [T][B] bit 0 terminated, bit 1 truncated; != 0 is the reference's `terminated or truncated` */
  double* final_observation;      /* [B][obs_dim] what step T would start from */
  int32_t* n_terminal;            /* [1] number of finished episodes in the rollout */
} gs_rollout_view;
int gs_rollout_download(gs_handle* h, const gs_rollout_view* out);
/* the same data where it lies, for a consumer on the GPU (valid until the next gs_rollout on the handle; waits for the
 * rollout to finish).  observations = obs_seq[0 .. T-1], next_observations = obs_seq[1 .. T] except for the n_terminal
 * transitions (t, b) = terminal_index[k], whose next observation is terminal_obs[k] (obs_seq[t + 1][b] being the fresh
 * observation after the reset). */
typedef struct gs_rollout_device {
  int32_t T, B, obs_dim, action_dim;
  const double* obs_seq;          /* [T + 1][B][obs_dim] */
  const double* actions;          /* [T][B][action_dim] */
  const double* rewards;          /* [T][B] */
  const uint8_t* terminals;       /* [T][B] */
  int32_t n_terminal, reserved;
  const int32_t* terminal_index;  /* [n_terminal][2] (t, b), in no particular order */
  const double* terminal_obs;     /* [n_terminal][obs_dim] */
} gs_rollout_device;
int gs_rollout_device_view(gs_handle* h, gs_rollout_device* out);

/* ---- checkpoint / resume: [B][state_dim] float64 blob ------------
 * layout per instance: time, step, constraint_violations, total_losses, episode_reward,
 * frequency, irradiance, wind, temperature, cloud, seed_lo, seed_hi,
 * soc[n_bats], battery_power[n_bats], curtailment[n_gens], Vm[n], Va[n], flow[m], loading[m] */
int gs_get_state(gs_handle* h, double* state);
int gs_set_state(gs_handle* h, const double* state);

/* ---- multi-GPU: one optional exchange per step, RCCL all-gather of observations over
 * xGMI.  RCCL is dlopen'ed on first use. --------------------------- */
int gs_comm_unique_id(uint8_t id_out[128]);
int gs_comm_init(gs_handle* h, const uint8_t id[128], int32_t rank, int32_t world_size);
/* gathers this handle's device-resident obs [B][obs_dim] from all ranks into a device
 * buffer [world*B][obs_dim]; obs_full_host may be NULL (stay on device) */
int gs_allgather_obs(gs_handle* h, double* obs_full_host);
int gs_comm_destroy(gs_handle* h);
/* What the communicator reports about this member: asked of RCCL itself (ncclCommCount, ncclCommUserRank, ncclCommCuDevice,
 * ncclGetVersion; -1 where the library lacks the entry point), plus the HIP device's UUID -- so that the output of an N-rank
 * run shows N ranks on N distinct devices.  transport: 1 RCCL, 2 the in-process loopback (nranks / rank from the group). */
typedef struct gs_comm_info_t {
  int32_t transport, nranks, rank, device, comm_device, rccl_version, reserved0, reserved1;
  uint8_t device_uuid[16];
} gs_comm_info_t;
int gs_comm_info(gs_handle* h, gs_comm_info_t* out);
/* The in-process transport: `nshards` handles of ONE process (shard r built with first_instance = r * B; normally all on
 * one device) form the communicator, rank = position in `shards`.  A member's compact block reaches the others by
 * device-to-device copies on their exchange streams where RCCL would move it over xGMI; compaction, slot offsets,
 * expansion, constant columns, the double-buffered observation buffers and their events are the code the RCCL transport
 * runs.  With this transport gs_allgather_obs completes when the LAST member has called (a grouped call): earlier callers
 * return at once, their obs_full_host (if any) is filled by that last call.  What it is for: rehearsing and testing the
 * N > 1 exchange on a box with fewer GPUs than ranks (tests/test_gpu_loopback.py; bench.py flags such a run
 * "transport": "loopback"), and a single process that keeps several shards on one GPU. */
int gs_comm_init_loopback(gs_handle* const* shards, int32_t nshards);
/* The form SURVEY.md section 8(b) specified: one call gathers for every member (all members of one loopback communicator,
 * or the RCCL ranks one process drives, issued as one ncclGroup); obs_full_host: NULL or [nshards * B][obs_dim], filled
 * from shard 0's gathered block (every member's is the same). */
int gs_allgather_obs_shards(gs_handle* const* shards, int32_t nshards, double* obs_full_host);
/* The gathered block where it lies, [world * B][obs_dim] in rank order, for a learner on the GPU (valid until the member's
 * next gather completes).  consumer_stream (hipStream_t as void*): made to wait on the device for the gather; NULL: the
 * call returns when the gather has finished.  gs_allgather_obs_download is the host copy of the same block. */
typedef struct gs_gathered_obs {
  const double* observations;
  int64_t rows;                   /* world * B */
  int32_t obs_dim, rank, world, reserved;
} gs_gathered_obs;
int gs_allgather_obs_view(gs_handle* h, gs_gathered_obs* out, void* consumer_stream);
int gs_allgather_obs_download(gs_handle* h, double* obs_full_host);

/* ---- measurement: HIP-event timing of every kernel launched on the handle's stream ------ */
enum { GS_K_UNPACK = 0, GS_K_ENV_PRE = 1, GS_K_SOLVE = 2, GS_K_ENV_POST = 3, GS_K_PACK = 4, GS_K_COUNT = 5 };
/* on = 1: a HIP event pair around every launch (per-kernel totals; the events themselves cost a few us per launch);
 * on = 2: ONE event pair around everything launched until the next gs_timing_read -- call that right after the last
 * launch of the region; it returns the span as the time of the most-launched kernel, i.e. duration + gaps per launch */
int gs_timing_enable(gs_handle* h, int32_t on);
/* diagnostic build aid: the first call arms per-phase cycle counters inside the solver kernels
 * (block 0 / wave 0; phases: prologue, init, mismatch, bottom-up, flag, top-down, final mismatch,
 * epilogue), later calls return the sums accumulated since the previous call and clear them. */
int gs_debug_stamps(gs_handle* h, uint64_t* cycles_out, int32_t n);
/* The host-side schedule of the meshed Newton-Raphson step kernel (gs_k_step_nr_mesh2; csrc/mesh_schedule.h) for a topology,
 * built WITHOUT a device: the block elimination that replaces np.linalg.solve (power_flow.py:186-190) as per-(wavefront, row,
 * sub-group) items, pull lists and Ybus rows.  header[16]: ok, n_levels, n_rows, max_rows_per_wave, n_pivots, msg_units,
 * n_messages, n_accumulators, max_degree, unit_bytes, zero_off, dummy_off, body_off, region_bytes, item_bytes, n_adj.  unit_budget:
 * the message units the level assignment tries to stay below (0: levels as early as possible).  Call once
 * with the arrays NULL for the sizes (items: nw * ni * 8 records of item_bytes; rowinfo: nw * ni * 4; adj_y: 2 * n_adj), then
 * with buffers.  tests/test_mesh_schedule.py replays the tables in NumPy against a dense solve. */
int gs_mesh_schedule_dump(const gs_topology* topo, int32_t zero_z_mode, int32_t nw, int32_t ni, int32_t acc_cap, int32_t unit_budget, int32_t region_base,
                          int32_t slot_bytes, int32_t* header, char* why, int32_t why_cap, void* items, int32_t* rowinfo,
                          int32_t* adj_off, double* adj_y);
/* The same schedule as the kernel reads it: 16 words per item (GS_MESH_W_*, csrc/gs_internal.h), rowinfo with the neighbour count
 * of the packed form, the Ybus table it stages in LDS ((n_pairs + 1) off-diagonal entries, then the diagonal entry of every
 * voltage slot) and the neighbour lists.  counts[4]: n_pairs, doubles of ytab, entries of adj_ent, words per item; arrays may be
 * NULL (first call).  GS_E_TOPOLOGY if the feeder is not eligible. */
/* Test aid, host arithmetic only: the constant linear map of the meshed member's first Newton step from the flat start,
 * x = W [P_spec of the non-slack buses in bus order; 1] (power_flow.py:125-134 flat start, :243-287 exact Jacobian, :186-190 solve; Q_spec = 0),
 * W row-major [2 (n - 1)][n]: rows (d theta, d|V|) per non-slack bus, last column the constant term.  All-PQ networks only. */
int gs_flat_newton_map_dump(const gs_topology* topo, int32_t zero_z_mode, double* out);
int gs_mesh_schedule_dump_packed(const gs_topology* topo, int32_t zero_z_mode, int32_t nw, int32_t ni, int32_t acc_cap, int32_t unit_budget, int32_t region_base,
                                 int32_t slot_bytes, int32_t* counts, int32_t* packed, int32_t* rowinfo, double* ytab, int32_t* adj_ent);
/* Diagnostic: (start, end) of each of the first n_blocks workgroups of the last step launch, in ticks of the GPU's 100 MHz
 * real-time clock (second-generation step kernels; arm with gs_debug_stamps while GS_STAMP_BLOCK_TIMES is set). */
int gs_debug_block_times(gs_handle* h, uint64_t* out, int32_t n_blocks);
/* total_ms[GS_K_COUNT], launches[GS_K_COUNT] accumulated since the last call; resets them */
int gs_timing_read(gs_handle* h, double* total_ms, int64_t* launches);
/* test aid: overwrite one family of device rows with values[B][width] (width = n, m or 1), so that kernels
 * reading the device state (the post-step checks below) can be driven with fixture data */
enum { GS_ROWS_VM = 0, GS_ROWS_LINE_LOADING = 1, GS_ROWS_ENV_LINE_LOADING = 2, GS_ROWS_LINE_FLOW = 3, GS_ROWS_FREQUENCY = 4,
       GS_ROWS_CONVERGED = 5, GS_ROWS_ITERATIONS = 6, GS_ROWS_MAX_MISMATCH = 7,
       GS_ROWS_LOAD_POWER = 8 /* [n_loads] realised load powers of the last step (dynamics.py:54-75) */, GS_ROWS_COUNT = 9 };
int gs_debug_write_rows(gs_handle* h, int32_t which, const double* values);
/* test aid: the same families read back as values[B][width] (what the statistical tests of the stochastic load
 * model look at: the realised load powers are not part of the observation, grid_env.py:769-770) */
int gs_debug_read_rows(gs_handle* h, int32_t which, double* values);

/* ======================================================================================
 * Post-step checks on the state a step / solve left on the device.
 * Replaces, batched and without a host round trip of voltages and loadings:
 *   SafetyChecker.check_constraints / is_safe / get_violation_severity   utils/safety.py:114-203
 *   SafetyMonitor.check_constraints                                      utils/safety.py:313-394
 *   AdvancedRobustPowerFlowSolver._assess_solution_quality               environments/robust_power_flow.py:615-657
 * A gs_checks object is bound to one gs_handle (it reads that handle's device state and runs on
 * its stream) and carries what the reference classes carry between calls: the previous voltages
 * and frequency (rate-of-change limits), the consecutive-violation counter and the sticky
 * emergency mode.  thermal_data is not modelled, so severity never reaches GS_SEVERITY_CRITICAL.
 * ====================================================================================== */
enum {  /* rows of gs_checks_view.ints, each [B]; C_ = SafetyChecker, M_ = SafetyMonitor */
  GS_CI_C_NLOW = 0, GS_CI_C_NHIGH, GS_CI_C_FLOW, GS_CI_C_FHIGH, GS_CI_C_NOVER, GS_CI_C_VRATE, GS_CI_C_FRATE, GS_CI_C_TOTAL,
  GS_CI_C_SEVERITY,   /* 0 safe, 1 low, 2 medium, 3 high, 4 critical (safety.py:188-203) */
  GS_CI_M_NHIGH, GS_CI_M_NLOW, GS_CI_M_NEMERG, GS_CI_M_FHIGH, GS_CI_M_FLOW, GS_CI_M_FEMERG, GS_CI_M_NOVER, GS_CI_M_TOTAL,
  GS_CI_M_ACTION,     /* emergency_action_required */
  GS_CI_M_CONSEC, GS_CI_M_EMODE, GS_CI_COUNT
};
enum { GS_CF_VRATE = 0, GS_CF_FRATE, GS_CF_QUALITY, GS_CF_COUNT };   /* rows of gs_checks_view.reals, each [B] */
enum {  /* bits of gs_checks_view.bus_mask[b][i] / line_mask[b][k] */
  GS_BM_C_LOW = 1, GS_BM_C_HIGH = 2, GS_BM_M_LOW = 4, GS_BM_M_HIGH = 8, GS_BM_M_EMERGENCY = 16,
  GS_LM_C_OVERLOAD = 1, GS_LM_M_OVERLOAD = 2
};

typedef struct gs_checks_config {
  int32_t struct_size;             /* = sizeof(gs_checks_config) */
  int32_t loading_source;          /* 0: |S|/rating of the load-flow solution (PowerFlowSolution.line_loadings);
                                      1: the environment's |P|/rating (Line.update_state, base.py:261-264) */
  double voltage_limits[2], frequency_limits[2], line_loading_limit;      /* SafetyChecker.__init__, safety.py:100-112 */
  double rate_voltage, rate_frequency, timestep;
  double mon_voltage_limits[2], mon_frequency_limits[2], mon_line_loading_limit;   /* SafetyMonitor.__init__, :296-311 */
  double mon_emergency_voltage[2], mon_emergency_frequency[2];
  double quality_tolerance;        /* the solver tolerance `_assess_solution_quality` compares max_mismatch with */
} gs_checks_config;

typedef struct gs_checks_view {    /* host buffers, any may be NULL */
  int32_t* ints;                   /* [GS_CI_COUNT][B] */
  double* reals;                   /* [GS_CF_COUNT][B] */
  uint8_t* bus_mask;               /* [B][n] */
  uint8_t* line_mask;              /* [B][m] */
} gs_checks_view;

typedef struct gs_checks gs_checks;

int gs_checks_create(gs_handle* h, const gs_checks_config* cfg, gs_checks** out);
void gs_checks_destroy(gs_checks* c);
/* frequency per instance for handles without an environment (solver-only); NULL = back to the handle's own */
int gs_checks_set_frequency(gs_checks* c, const double* frequency_hz);
/* one check_constraints call per instance on the handle's current device state (asynchronous) */
int gs_checks_run(gs_checks* c);
int gs_checks_download(gs_checks* c, const gs_checks_view* out);
/* forget previous state, counters and emergency mode of the masked instances (NULL = all): a freshly
 * constructed SafetyChecker() / SafetyMonitor() */
int gs_checks_reset(gs_checks* c, const uint8_t* mask);
/* HIP-event timing of gs_checks_run is off unless enabled (a per-step check over a long run records nothing) */
int gs_checks_timing_enable(gs_checks* c, int32_t on);
int gs_checks_timing_read(gs_checks* c, double* total_ms, int64_t* launches);
/* on != 0: every later gs_step / gs_step_device evaluates these checks inside its own kernel (the epilogue already holds
 * the voltages and loadings), exactly as one gs_checks_run after the step would; gs_checks_download then returns the
 * checks of the last step.  Do not call gs_checks_run as well (the stateful parts would advance twice).  One fused checks
 * object per handle; the environment's |P| / rating or the solution's |S| / rating as configured. */
int gs_checks_set_fused(gs_checks* c, int32_t on, int32_t want_masks);

/* ======================================================================================
 * Three-phase unbalanced radial load flow (BASELINE.json config 5).  NEW functionality: the
 * reference names UnbalancedPowerFlow (README.md:187-197, API_REFERENCE.md:420) but contains no
 * implementation, so there is no reference interface to cite beyond the solver plug point
 * (environments/power_flow.py:38-46) whose record layout the outputs follow, with a phase axis.
 * Mapping differs from the single-phase path: one workgroup per instance (networks of thousands
 * of nodes, batches of ~1000).  Feeders of up to ~9 700 phase conductors stay in that workgroup's
 * registers and LDS for the whole solve, the sweeps evaluated as prefix sums over depth-first
 * orders of the tree (csrc/gridstep3_resident.h); larger ones are swept level by level through HBM.
 * ====================================================================================== */
typedef struct gs3_topology {
  int32_t struct_size;            /* = sizeof(gs3_topology) */
  int32_t n;                      /* nodes; node `source` is the three-phase source (slack) */
  int32_t source;
  int32_t reserved;
  const int32_t* parent;          /* [n] upstream node of each node, -1 for the source */
  const uint8_t* phases;          /* [n] bit mask (1 = a, 2 = b, 4 = c) of the phases the node's upstream line carries;
                                     must be a subset of the parent's mask; the source has 7 */
  const double* z_re;             /* [n][3][3] series impedance of the upstream line, per unit (ignored for the source) */
  const double* z_im;             /* [n][3][3] */
  const double* v_source;         /* [3] source voltage magnitudes; angles are 0, -120, +120 degrees */
} gs3_topology;

typedef struct gs3_solution_view {
  double* v_re;                   /* [B][n][3] phase-to-neutral voltage, rectangular; absent phases = 0 */
  double* v_im;                   /* [B][n][3] */
  double* losses;                 /* [B] total real losses (sum over phases) */
  double* max_mismatch;           /* [B] */
  int32_t* iterations;            /* [B] */
  uint8_t* converged;             /* [B] */
} gs3_solution_view;

typedef struct gs3_handle gs3_handle;

int gs3_create(const gs3_topology* topo, double tolerance, int32_t max_iterations, int32_t batch, int32_t device,
               gs3_handle** out);
void gs3_destroy(gs3_handle* h);
const char* gs3_last_error(const gs3_handle* h);
/* P_spec / Q_spec [B][n][3]: net injection per node and phase (generation - load), per unit */
int gs3_solve(gs3_handle* h, const double* P_spec, const double* Q_spec, const gs3_solution_view* out);
/* device-resident variant for measurement */
int gs3_upload_injections(gs3_handle* h, const double* P_spec, const double* Q_spec);
int gs3_solve_device(gs3_handle* h);
int gs3_download_solution(gs3_handle* h, const gs3_solution_view* out);
int gs3_synchronize(gs3_handle* h);
/* average milliseconds of the solve kernel over the launches since the last call (HIP events) */
int gs3_timing_read(gs3_handle* h, double* total_ms, int64_t* launches);
int gs3_describe(const gs3_handle* h, char* buf, int32_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* GRIDSTEP_H */
