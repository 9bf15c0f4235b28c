/* robogym.h -- C ABI of librobogym_hip.so: the MI355X-native (gfx950) vectorised
 * Robotarium-gym step engine.
 *
 * Drop-in boundary.  Each entry point names the reference interface
 * it replaces; paths are relative to /root/reference/robotarium_gym/.  The reference is pure
 * Python and has no FFI of its own: the binding a maintainer would add is the ctypes layer in
 * marbler_amd/_lib.py (shown in INTEGRATION.md).
 *
 * Conventions
 *  - Every pointer in rg_state / rg_step_io / rg_step arguments is a DEVICE pointer into memory
 *    owned by the caller (torch-ROCm tensors: `tensor.data_ptr()`).  The library never allocates
 *    or frees caller-visible memory.
 *  - All calls are asynchronous on the HIP stream given to rg_create; no call synchronises.
 *  - Every call on a handle runs with the handle's device current (hipSetDevice(device of rg_create))
 *    and restores the caller's current device before returning: a handle created for GPU 1 launches
 *    on GPU 1 whatever device the calling thread has selected.  The stream must belong to that device.
 *  - Return value: 0 = OK, negative = error (text via rg_last_error()).  No exceptions cross
 *    the ABI.  A handle is re-entrant across handles, not thread-safe within one.
 *  - E envs, N agents per env, P prey, D per-agent observation length.
 */
#ifndef ROBOGYM_H
#define ROBOGYM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 5): rg_scenario_params ends in the barrier-QP solver selection (qp_mode + cvxopt's options): RG_QP_CVXOPT computes the
 * interior-point iterate the reference's stack computes (utilities/controller.py:13-16,23) instead of the exact projection;
 * rg_step_io.zero_obs_on_end; + rg_actor_forward_explore(), rg_actor_pack_gru_f16x2() and rg_actor_weights.gru_packed == 3.
 * 5 (round 4): + rg_actor_pack_gru_bf16x3() and rg_actor_weights.gru_packed == 2; rg_rollout takes every shape (no E*N*D % 4
 * rule); the one-lane-per-env step kernel covers N <= 6. */
#define RG_ABI_VERSION 6
#define RG_MAX_AGENTS 16
#define RG_MAX_PREY 64

/* scenarios/<S>/ : wrapper.py:12-16 env_dict */
enum {
    RG_SCN_PREDATOR_CAPTURE_PREY = 0,
    RG_SCN_WAREHOUSE = 1,
    RG_SCN_MATERIAL_TRANSPORT = 2,
    RG_SCN_SIMPLE = 3,
    RG_SCN_ARCTIC_TRANSPORT = 4
};
#define RG_ARCTIC_ROWS 8
#define RG_ARCTIC_COLS 12
/* rps _validate collision test */
enum { RG_COLLISION_CENTER = 0, RG_COLLISION_OFFSET = 1 };
/* info['message'] of the reference (utilities/roboEnv.py:82-94) as a code */
enum { RG_VIOL_NONE = 0, RG_VIOL_COLLISION = 1, RG_VIOL_BOUNDARY = 2, RG_VIOL_COLLISION_BOUNDARY = 3 };
/* How `si_barrier_cert` (utilities/controller.py:13-16,23: rps' certificate closure -> cvxopt `qp`) is evaluated:
 * RG_QP_EXACT  -- the exact solution of the QP (the Euclidean projection), by Hildreth sweeps (qp_rtol, qp_max_sweeps);
 * RG_QP_CVXOPT -- the iterate cvxopt's interior-point method stops at under rps' options (reltol = feastol = 1e-2, maxiters 50,
 *                 abstol 1e-7): an approximate, strictly interior solution -- what the reference's own stack returns.  Restated
 *                 coneqp in binary64 (csrc/ipm_qp.h); n_agents <= 8. */
enum { RG_QP_EXACT = 0, RG_QP_CVXOPT = 1 };
#define RG_QP_CVXOPT_MAX_AGENTS 8

/* Reset geometry of rps generate_initial_conditions (Appendix A.7) + the scenario's shift
 * (misc.py:49-63, warehouse.py:93-98): N distinct cells of an nx x ny grid;
 *   x = ((cx*spacing - w2) + ox1) + ox2,  y = ((cy*spacing - h2) + oy1) + oy2.
 * nx, ny are computed by the host in float64 exactly as the reference does
 * (floor(width/spacing), where width itself is a float64 difference of YAML values). */
typedef struct rg_grid {
    int32_t nx, ny;
    float spacing, w2, h2, ox1, ox2, oy1, oy2;
} rg_grid;

/* The scenario config.yaml (scenarios/<S>/config.yaml, read by wrapper.py:27-31) plus the rps
 * constants (Appendix A), flattened.  This is the block broadcast to every GPU at init. */
typedef struct rg_scenario_params {
    int32_t scenario;
    int32_t n_agents;
    int32_t obs_dim;                 /* D */
    int32_t update_frequency;        /* U, roboEnv.py:52 */
    int32_t controller_period;       /* 15, roboEnv.py:63 */
    int32_t max_episode_steps;
    int32_t penalize_violations;     /* roboEnv.py:82 */
    int32_t barrier_has_unsafe_gain; /* 1 = certificate2 ('safe'), 0 = certificate ('default'), controller.py:13-16 */
    int32_t collision_variant;       /* RG_COLLISION_* */
    int32_t capability_aware;
    int32_t num_prey;                /* P */
    int32_t num_neighbors;           /* K */
    int32_t torque[RG_MAX_AGENTS];   /* MaterialTransport.py:72-76 */
    /* rps constants */
    float time_step, bound_x0, bound_y0, bound_w, bound_h;
    float robot_diameter, wheel_radius, max_linear_velocity;
    float collision_offset, collision_diameter;
    float projection_distance, angular_velocity_limit, position_velocity_limit;
    float barrier_gain, unsafe_barrier_gain, safety_radius, barrier_magnitude_limit;
    /* barrier QP solver (sim_spec_v0: Hildreth sweeps): stop when the largest change of a sweep is
     * <= qp_rtol * max(|u|_inf, barrier_magnitude_limit), or after qp_max_sweeps sweeps */
    float qp_rtol;
    int32_t qp_max_sweeps;
    /* scenario */
    float left, right, up, down;
    float agent_step[RG_MAX_AGENTS];     /* step_dist, or MaterialTransport per-agent speed */
    float sensing_radius[RG_MAX_AGENTS]; /* PredatorCapturePrey.py:40-44 */
    float capture_radius[RG_MAX_AGENTS];
    float time_penalty, sense_reward, capture_reward, violation_reward;
    float load_reward, unload_reward, goal_width;
    float unload_multiplier, load_multiplier, end_goal_width, zone1_radius;
    float reward_scaler;                          /* Simple (simple.py:219-222) */
    float arctic_normal_step, arctic_slow_step, arctic_fast_step; /* ArcticTransport agent.py:89-113 */
    float not_reached_penalty, dist_multiplier;   /* ArcticTransport.py:125-134 */
    /* reset (misc.py:49-63, scenario reset()) */
    rg_grid agent_grid, prey_grid;
    int32_t keep_theta;              /* Warehouse keeps the sampled heading, misc.py:58,62 zero it */
    int32_t shared_reward;           /* config key: episode return adds reward[0] (1) or sum(reward) (0), misc.py:178-181 */
    float zone1_mean, zone1_std, zone2_mean, zone2_std; /* MaterialTransport.py:99-100 */
    /* barrier QP solver (ABI 6): RG_QP_*; the ipm_* fields are cvxopt's `solvers.options` as rps sets them at import
     * (abstol 1e-7 = cvxopt's default, reltol = feastol = 1e-2, maxiters 50) and are read in RG_QP_CVXOPT mode only */
    int32_t qp_mode;
    float ipm_abstol, ipm_reltol, ipm_feastol;
    int32_t ipm_maxiters;
} rg_scenario_params;

/* Env state in HBM (replaces the Python objects: scenario.agent_poses 3xN = the live alias of
 * rps.Robotarium.poses, roboEnv.previous_pose, scenario.episode_steps, prey / loaded / load /
 * zone / message attributes).  Arrays a scenario does not use may be NULL. */
typedef struct rg_state {
    float *poses;           /* [E][3][N]: x row, y row, theta row, as the reference's 3xN */
    float *carry_dist;      /* [E][N]: length of the last sub-iteration of the previous step, not yet
                               added to dist_travelled (roboEnv.py:55-59 lags one iteration) */
    int32_t *episode_steps; /* [E] */
    int32_t *reset_count;   /* [E] episodes started so far (RNG stream position) */
    float *prey_loc;        /* [E][P][2]   PredatorCapturePrey; Simple keeps its goal here (P = 1) */
    uint8_t *prey_sensed;   /* [E][P] */
    uint8_t *prey_captured; /* [E][P] */
    uint8_t *loaded;        /* [E][N]      Warehouse */
    int32_t *load;          /* [E][N]      MaterialTransport */
    int32_t *zone_load;     /* [E][2] */
    int32_t *messages;      /* [E][4] */
    uint8_t *grid;          /* [E][8*12]   ArcticTransport terrain: 0 normal, 1 ice, 2 water, 3 goal */
    int32_t *goal_col;      /* [E]         ArcticTransport goal_loc[1] (goal_loc[0] is always 1) */
    uint8_t *pixel_type;    /* [E][N]      ArcticTransport Agent.pixel_type (read by the NEXT step's goals) */
    uint8_t *reached_goal;  /* [E][N]      ArcticTransport Agent.reached_goal */
    /* Optional rollout statistics (all four NULL, or all four set), the on-device form of the
     * accumulators in utilities/misc.py:151-206 (episodeReward, episodeSteps, totalReward): */
    float *ep_return;       /* [E] return of the running episode */
    float *done_return_sum; /* [E] sum of the returns of the episodes this env has finished */
    int32_t *done_count;    /* [E] number of episodes this env has finished */
    int32_t *done_steps_sum;/* [E] total length of those episodes */
    /* Optional scratch of the lane-group step kernel (both NULL, or both set; contents are derived data, never part of a
     * snapshot): an env's NEXT initial state, drawn ahead of time at the end of a launch in which the env did not finish,
     * so that the launch in which it does finish copies it instead of running the sampler on its critical path.
     * next_init: [E][rg_next_init_stride(params)] floats; next_episode: [E], the episode index (= reset_count) the block
     * was drawn for, -1 = none.  The library marks every block stale itself (next_episode <- -1, one memset on the
     * handle's stream) on the first rg_step / rg_rollout after rg_bind_state and whenever their `seed` argument changes;
     * a caller that rewrites state arrays behind the handle's back (restoring a snapshot) resets next_episode to -1. */
    float *next_init;
    int32_t *next_episode;
} rg_state;

/* Everything Wrapper.step returns (wrapper.py:41-44), batched. */
typedef struct rg_step_io {
    float *obs;            /* [E][N][D] */
    float *reward;         /* [E][N] */
    uint8_t *done;         /* [E]  (the reference repeats it N times) */
    float *dist_travelled; /* [E][N]  info['dist_travelled'] */
    uint8_t *violation;    /* [E]     info['message'] as RG_VIOL_* */
    int32_t *remaining;    /* [E]     info['remaining'], -1 when the reference omits the key */
    int32_t *qp_sweeps;    /* [E] or NULL: diagnostic, max barrier-QP sweeps in this step */
    /* Optional (elapsed NULL = off): what EPyMARL's gymma wrapper puts around the env -- gym's TimeLimit and the
     * reductions float(sum(reward_n)), all(done_n) (README.md:25-28 of the reference; the wrapper itself is external) --
     * computed by the same launch instead of a dozen elementwise launches around it. */
    int32_t *elapsed;      /* [E] in/out (state, not per step): steps since the env's last (re)start, TimeLimit's counter */
    uint8_t *truncated;    /* [E] out: 1 = the time limit, not the scenario, ended the episode in this step (info['TimeLimit.truncated']) */
    uint8_t *ended;        /* [E] out: done | truncated -- what gymma returns as `terminated`; with auto_reset such an env is
                              reset in the same launch, and a truncated episode is booked in the statistics like a finished one */
    float *reward_sum;     /* [E] out: the sum of the agents' rewards in agent order */
    int32_t time_limit;    /* TimeLimit's max_episode_steps (> 0) */
    int32_t zero_obs_on_end; /* (with the gymma block) nonzero: the observation rows of an env that ENDS in this step are written as
                              zeros -- gymma users see the reset observation (what the reference's reset() returns,
                              PredatorCapturePrey.py:136) as the next observation of an episode that just ended; `obs` may then
                              point straight into the trainer's [T + 1][E][N][D] batch, slot t + 1 */
} rg_step_io;

typedef struct rg_handle rg_handle;

int rg_abi_version(void);
const char *rg_last_error(void);
/* sizeof(rg_scenario_params / rg_state / rg_step_io) as compiled, for binding self-checks */
int rg_sizeof_params(void);
int rg_sizeof_state(void);
int rg_sizeof_step_io(void);
/* floats per env of rg_state.next_init for this parameter block */
int rg_next_init_stride(const rg_scenario_params *params);

/* Replaces Wrapper.__init__ -> scenario.__init__ -> roboEnv.__init__ -> Controller.__init__
 * (wrapper.py:20-34, PredatorCapturePrey.py:15-59, roboEnv.py:12-24, controller.py:5-18).
 * `env_offset` is the global index of this shard's env 0 (RNG streams are keyed by global env
 * index, so results do not depend on how envs are sharded over GPUs). */
rg_handle *rg_create(const rg_scenario_params *params, int32_t num_envs, int64_t env_offset, int32_t device,
                     void *hip_stream);
int rg_destroy(rg_handle *h);

/* Binds the caller-owned state arrays.  Must be called before reset/step. */
int rg_bind_state(rg_handle *h, const rg_state *state);

/* Replaces scenario.reset() + roboEnv.reset() (PredatorCapturePrey.py:114-136, warehouse.py:84-100,
 * MaterialTransport.py:94-111, roboEnv.py:27-36,98-118, misc.py:49-63).  mask: [E] uint8 device
 * pointer, nonzero = reset that env; NULL = all.
 * The running return of a reset env (rg_state.ep_return) restarts at zero.  flags:
 * RG_RESET_BOOK_EPISODE -- the abandoned episode (if it has taken at least one step) is first added to
 * done_return_sum / done_count / done_steps_sum, as an episode that a time limit outside the scenario
 * cut short (gym's TimeLimit in EPyMARL's gymma wrapper; `run_env` counts such episodes, misc.py:186-206). */
#define RG_RESET_BOOK_EPISODE 1
int rg_reset(rg_handle *h, const uint8_t *mask, uint64_t seed, int32_t flags);

/* Replaces Wrapper.step -> scenario.step -> roboEnv.step -> Controller.set_velocities ->
 * rps.Robotarium.{get_poses,set_velocities,step} and the scenario's tracking / observation /
 * reward / termination code (wrapper.py:41-44, PredatorCapturePrey.py:138-216, warehouse.py:102-178,
 * MaterialTransport.py:113-189, roboEnv.py:38-96, controller.py:20-24).
 * actions: [E][N] int32.  If auto_reset != 0, envs that finish are reset in the same launch
 * (their returned obs/reward/done are those of the terminal step). */
int rg_step(rg_handle *h, const int32_t *actions, const rg_step_io *io, int32_t auto_reset, uint64_t seed);

/* num_steps consecutive rg_step()s in ONE launch, for action sequences that are known up front
 * (random-policy rollouts -- misc.py:134-221 `run_env` with a random policy --, replayed logs,
 * open-loop plans).  actions: [K][E][N]; every array of io has a leading dimension K and receives
 * what the k-th rg_step would have returned; results are bit-identical to K calls of rg_step.
 * Envs advance independently inside the launch (no device-wide synchronisation between steps), so
 * the launch takes K mean steps, not K worst-case steps. */
int rg_rollout(rg_handle *h, const int32_t *actions, int32_t num_steps, const rg_step_io *io, int32_t auto_reset,
               uint64_t seed);

/* The HIP stream later launches of this handle go to (rg_create's stream until changed).  Lets the
 * caller record rg_step into a hipGraph: set the capturing stream, capture, replay -- the evaluation
 * loop of misc.py:155-185 (policy forward, arg-max, env step) becomes one graph launch per step. */
int rg_set_stream(rg_handle *h, void *hip_stream);

/* The observation the scenario would build from the current state without stepping (the
 * reference returns zeros from reset(), PredatorCapturePrey.py:136; EPyMARL's gymma layer is
 * where get_obs() lives).  obs: [E][N][D]. */
int rg_get_obs(rg_handle *h, float *obs);

/* Which step kernel this handle launches: 0 = lane group per env (small / medium batches, and N >= 7 at every batch size:
 * the one-lane-per-env kernel is instantiated for N <= 6 only since round 4), 1 = one lane per env (chip-filling batches).  Chosen in rg_create from (scenario, n_agents, num_envs) -- measured cross-overs, csrc/robogym_capi.hip
 * tpe_min_envs -- or forced by the environment variable RG_STEP_KERNEL=group|tpe.  Both give bit-identical results; the
 * query exists so that tests and profiles can say which one ran.  Negative: error. */
int rg_step_kernel(const rg_handle *h);

/* ---- policy inference for evaluation rollouts ------------------------
 * The EPyMARL recurrent actor the reference evaluates with (utilities/rnn_agent.py:5-29 `RNNAgent`:
 * fc1 -> ReLU -> GRUCell -> fc2; utilities/rnn_ns_agent.py:5-36 `RNNNSAgent`: one per agent), for all
 * E x N agents in one launch on the matrix cores (f32-input MFMA: float32 products and sums).
 * Weight arrays are torch's parameter layouts ([out][in] row-major), stacked over n_sets = 1 (shared)
 * or N (one set per agent).  use_rnn = 0 (rnn_agent.py:13,27: Linear + ReLU instead of the GRU): that
 * layer's weight / bias go in wih / bih, whh / bhh are ignored. */
typedef struct {
    const float *w1, *b1;   /* [S][H][I], [S][H] */
    const float *wih, *bih; /* [S][3H][H], [S][3H]   (use_rnn = 0: [S][H][H], [S][H]) */
    const float *whh, *bhh; /* [S][3H][H], [S][3H] */
    const float *w2, *b2;   /* [S][A][H], [S][A] */
    int32_t n_sets, input_dim, hidden_dim, n_actions, use_rnn;
    int32_t gru_packed;     /* 0: wih / whh in torch's layout; 1: the float32 streaming order written by rg_actor_pack_gru;
                               2: three bfloat16 planes written by rg_actor_pack_gru_bf16x3 (6 bytes per weight);
                               3: two binary16 planes written by rg_actor_pack_gru_f16x2 (4 bytes per weight) */
} rg_actor_weights;

/* One actor step (misc.py:160-170: `actor(obs, hs)` then arg-max).  obs [E][N][D]; with
 * append_agent_id the one-hot agent id is appended to each row (misc.py:162-164), D (+N) must equal
 * input_dim.  hidden [E][N][H] is updated in place; restart [E] (or NULL) nonzero = a new episode in that
 * env: its hidden state and its observation are taken as zero (what the reference's reset() returns,
 * PredatorCapturePrey.py:136) -- pass the done flags of the previous rg_step.  q [E][N][A] (or NULL) receives the action values,
 * actions [E][N] (or NULL) the greedy action.  hidden_dim 64 or 128, n_actions <= 32, input_dim <= 64. */
int rg_actor_forward(const rg_actor_weights *w, int32_t num_envs, int32_t n_agents, const float *obs,
                     int32_t obs_dim, int32_t append_agent_id, const uint8_t *restart, float *hidden, float *q,
                     int32_t *actions, void *hip_stream);
/* The same launch with the epsilon-greedy selection of EPyMARL's action selector (components/action_selectors.py, external to
 * the reference: its runners explore this way during training) folded in: explore_u [E][N] holds one uniform draw in [0, 1) per
 * agent; with k = (int)(u * (n_actions / epsilon)) in binary32, the action is k when k < n_actions (that is u < epsilon, and k is
 * then uniform over the actions) and the greedy action otherwise.  epsilon in [1e-6, 1]; actions must not be NULL. */
int rg_actor_forward_explore(const rg_actor_weights *w, int32_t num_envs, int32_t n_agents, const float *obs,
                             int32_t obs_dim, int32_t append_agent_id, const uint8_t *restart, float *hidden, float *q,
                             int32_t *actions, const float *explore_u, float epsilon, void *hip_stream);
/* Optional, once per actor: reorder a GRU weight array ([S][3H][H], torch layout) into the order the
 * kernel streams it (1 KB per load instruction instead of 64 scattered 16-byte pieces).  dst: a device
 * buffer of the same size; use it as wih / whh with gru_packed = 1. */
int rg_actor_pack_gru(const float *src, int32_t n_sets, int32_t hidden_dim, float *dst, void *hip_stream);
/* The same matrix as three bfloat16 planes -- w = hi + mid + lo exactly up to 2^-24 |w|, each plane the top 16 bits of what
 * the planes before it leave -- in the order the kernel streams them.  dst: 6 bytes per weight = 3/2 of the size of src,
 * 16-byte aligned; use it as wih / whh with gru_packed = 2.  The GRU's products then run on the bfloat16 matrix cores (16 x
 * the float32 MFMA rate) as six plane products per float32 product, the hidden state and fc1's output being split the same
 * way on the fly; the three products of order 2^-24 are left out, which keeps the error below the rounding error of a
 * float32 dot product of the same length (rnn_agent.py:24 `self.rnn(x, h_in)` evaluated in float32 is the reference). */
int rg_actor_pack_gru_bf16x3(const float *src, int32_t n_sets, int32_t hidden_dim, void *dst, void *hip_stream);
/* The same matrix as TWO binary16 planes -- hi = binary16(w), lo' = binary16(2^11 (w - hi)): w = hi + 2^-11 lo' up to 2^-22 |w|
 * -- in the order the kernel streams them.  dst: 4 bytes per weight = the size of src, 16-byte aligned; use it as wih / whh with
 * gru_packed = 3.  A float32 product is then THREE plane products on the binary16 matrix cores (hi hi into one accumulator, the
 * two cross products into a second that joins it scaled by 2^-11), the hidden state and fc1's output being split the same way on
 * the fly: half the matrix-core time of the three-plane form, error at the level of a float32 GEMM's own roundings (measured
 * max 3.9e-7 against 9.5e-7 on 128-long dot products).  Range: binary16's -- activations above 65 504 saturate (the hidden state
 * is in [-1, 1], fc1's output a ReLU of the observation's affine image); below 2^-14 the first plane is a binary16 denormal, which
 * the matrix cores take as it is. */
int rg_actor_pack_gru_f16x2(const float *src, int32_t n_sets, int32_t hidden_dim, void *dst, void *hip_stream);
const char *rg_actor_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ROBOGYM_H */
