// robogym_capi.hip -- the C ABI declared in include/robogym.h (host side).
//
// Plain pointers and sizes in, status codes out; no torch types.  The handle owns nothing on
// the device: state and outputs live in caller-owned HBM (torch-ROCm tensors).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "../../include/robogym.h"
#include "kernel_args.h"

struct rg_handle {
    rg_scenario_params params;
    rg::Consts consts;
    rg_state state;
    int32_t num_envs;
    int64_t env_offset;
    int32_t device;
    hipStream_t stream;
    bool bound;
    bool use_tpe;  // step with the thread-per-env kernel (robogym_tpe.hip) instead of the lane-group kernel
    bool seed_seen;      // the precomputed-reset blocks (rg_state.next_init) were drawn with last_seed
    uint64_t last_seed;
};

// Which step kernel: both give identical results.  The lane-group kernel has the shorter chain for
// small batches and is linear in the batch; the thread-per-env kernel's time is a step function of how many
// generations of wavefronts the batch needs (65 536 envs per generation and wave slot per SIMD), so it takes
// over where one of its steps undercuts the line.  Measured cross-overs on MI355X, final kernels of round 3
// (tools/crossover_probe.py, profiles/r3_crossover_probe.txt): the sparse collision pre-test made
// the lane-group kernel 5-8 % faster at these batch sizes and moved every threshold up from round 2's 53 248 / 65 536;
// the one-division restart of the barrier QP then favoured the lane-group kernel at N = 6, where the thread-per-env kernel
// runs on spilled registers, and compiling the thread-per-env files without the SLP vectoriser (build.py FILE_FLAGS: 57 fewer
// spilled values at N = 6, -22 %) gave most of that back: the table below is the last measurement of round 3.
// For N >= 7 the per-lane register footprint (28 pairs) leaves one wave per SIMD and the lane-group kernel -- at
// 92 % VALU issue there -- stays ahead at every batch size.  RG_STEP_KERNEL=group|tpe forces one (tests, profiling).
static int32_t tpe_min_envs(const rg_scenario_params &p) {
    const bool mt = p.scenario == RG_SCN_MATERIAL_TRANSPORT, pcp = p.scenario == RG_SCN_PREDATOR_CAPTURE_PREY;
    // RG_QP_CVXOPT: the launch is the interior-point iteration, one wave per SIMD in either kernel.  A lane group carries one env
    // through it in ~0.7 x the time a single lane needs, a wavefront of lanes carries eight times as many: the lane-group kernel
    // while its waves fit the chip at once (8 192 envs), the other one from the second generation on.  Measured ladder, round 5
    // (tools/ipm_probe.py --cross, profiles/r5_ipm_crossover.jsonl): N = 5  8 192 envs 166 vs 211 us, 16 384 290 vs 215;
    // N = 4  16 384 95 vs 95, 24 576 157 vs 99; at 65 536 x 5 981 vs 260 us.
    if (p.qp_mode == RG_QP_CVXOPT) return p.n_agents == 5 ? 12288 : 20480;
    switch (p.n_agents) {
        case 2: return 65536;
        case 3: return pcp ? 98304 : 65536;
        case 4: return p.scenario == RG_SCN_SIMPLE ? 393216 : p.scenario == RG_SCN_ARCTIC_TRANSPORT ? 131072 : 196608;
        case 5: return mt ? 49152 : 65536;
        case 6: return mt ? 65536 : p.scenario == RG_SCN_WAREHOUSE ? 262144 : pcp ? 98304 : 131072;
        default: return INT32_MAX;
    }
}

static thread_local char g_err[512] = "";

// The handle's device is current for the duration of a call; the caller's device is restored on return.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = (err == hipSuccess);
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

static int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

static int check_params(const rg_scenario_params *p) {
    if (!p) return fail(-1, "params is NULL");
    if (p->scenario < RG_SCN_PREDATOR_CAPTURE_PREY || p->scenario > RG_SCN_ARCTIC_TRANSPORT)
        return fail(-2, "unknown scenario id");
    if (p->n_agents < 1 || p->n_agents > RG_MAX_AGENTS) return fail(-3, "n_agents must be in 1..16");
    if (p->update_frequency < 1 || p->controller_period < 1) return fail(-4, "update_frequency / controller_period < 1");
    if (p->obs_dim < 1) return fail(-5, "obs_dim < 1");
    if (p->qp_max_sweeps < 1 || !(p->qp_rtol >= 0.0f)) return fail(-5, "qp_max_sweeps < 1 or qp_rtol < 0");
    if (p->qp_mode != RG_QP_EXACT && p->qp_mode != RG_QP_CVXOPT) return fail(-5, "unknown qp_mode");
    if (p->qp_mode == RG_QP_CVXOPT) {
        if (p->n_agents > RG_QP_CVXOPT_MAX_AGENTS) return fail(-3, "qp_mode RG_QP_CVXOPT is built for n_agents <= 8");
        if (p->ipm_maxiters < 0 || !(p->ipm_reltol >= 0.0f) || !(p->ipm_feastol > 0.0f) || !(p->ipm_abstol >= 0.0f))
            return fail(-5, "ipm_maxiters < 0, or a negative / zero cvxopt tolerance");
    }
    if (p->collision_variant != RG_COLLISION_CENTER && p->collision_variant != RG_COLLISION_OFFSET)
        return fail(-6, "unknown collision_variant");
    const rg_grid &g = p->agent_grid;
    // rps generate_initial_conditions asserts cells > N (Appendix A.7)
    if (g.nx < 1 || g.ny < 1 || g.nx * g.ny <= p->n_agents || g.nx * g.ny > 64)
        return fail(-7, "agent reset grid must have n_agents < nx*ny <= 64");
    if (p->scenario == RG_SCN_PREDATOR_CAPTURE_PREY) {
        if (p->num_prey < 1 || p->num_prey > RG_MAX_PREY) return fail(-8, "num_prey must be in 1..64");
        const rg_grid &q = p->prey_grid;
        if (q.nx < 1 || q.ny < 1 || q.nx * q.ny <= p->num_prey || q.nx * q.ny > 64)
            return fail(-9, "prey reset grid must have num_prey < nx*ny <= 64");
        const int od = p->capability_aware ? 6 : 4;
        const int nb = p->num_neighbors >= p->n_agents - 1 ? p->n_agents - 1 : p->num_neighbors;
        if (p->obs_dim < od * (nb + 1)) return fail(-10, "obs_dim too small for PredatorCapturePrey");
        // the 4-float observation blocks are written with 16-byte stores
        if (od == 4 && (p->obs_dim & 3)) return fail(-10, "obs_dim must be a multiple of 4 for PredatorCapturePrey without capability_aware");
    } else if (p->scenario == RG_SCN_WAREHOUSE) {
        const int nb = p->num_neighbors >= p->n_agents - 1 ? p->n_agents - 1 : p->num_neighbors;
        if (p->obs_dim < 3 * (nb + 1)) return fail(-10, "obs_dim too small for Warehouse");
    } else if (p->scenario == RG_SCN_SIMPLE) {
        if (p->num_prey != 1) return fail(-8, "Simple has one goal (num_prey = 1)");
        const rg_grid &q = p->prey_grid;
        if (q.nx < 1 || q.ny < 1 || q.nx * q.ny <= 1 || q.nx * q.ny > 64) return fail(-9, "goal reset grid must have 1 < nx*ny <= 64");
        if (p->obs_dim < 2 * (p->n_agents + 1)) return fail(-10, "obs_dim too small for Simple");
    } else if (p->scenario == RG_SCN_ARCTIC_TRANSPORT) {
        if (p->n_agents != 4) return fail(-3, "ArcticTransport has exactly 4 agents");
        if (p->obs_dim < 30) return fail(-10, "obs_dim too small for ArcticTransport");
    } else {
        if (p->obs_dim < (p->capability_aware ? 11 : 9)) return fail(-10, "obs_dim too small for MaterialTransport");
    }
    return 0;
}

extern "C" {

int rg_abi_version(void) { return RG_ABI_VERSION; }

const char *rg_last_error(void) { return g_err; }

int rg_sizeof_params(void) { return static_cast<int>(sizeof(rg_scenario_params)); }
int rg_sizeof_state(void) { return static_cast<int>(sizeof(rg_state)); }
int rg_sizeof_step_io(void) { return static_cast<int>(sizeof(rg_step_io)); }
int rg_next_init_stride(const rg_scenario_params *params) {
    if (check_params(params) != 0) return -1;
    return rg::next_init_stride(*params);
}

rg_handle *rg_create(const rg_scenario_params *params, int32_t num_envs, int64_t env_offset, int32_t device,
                     void *hip_stream) {
    if (check_params(params) != 0) return nullptr;
    if (num_envs < 1) {
        fail(-11, "num_envs < 1");
        return nullptr;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
        fail(-12, "no HIP device visible: librobogym_hip has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= count) {
        fail(-13, "device index out of range");
        return nullptr;
    }
    rg_handle *h = new (std::nothrow) rg_handle();
    if (!h) {
        fail(-14, "out of host memory");
        return nullptr;
    }
    h->params = *params;
    h->consts = rg::make_consts(*params);
    memset(&h->state, 0, sizeof(h->state));
    h->num_envs = num_envs;
    h->env_offset = env_offset;
    h->device = device;
    h->stream = static_cast<hipStream_t>(hip_stream);
    h->bound = false;
    h->seed_seen = false;
    h->last_seed = 0;
    h->use_tpe = rg::tpe_supported(*params) && num_envs >= tpe_min_envs(*params);
    if (const char *force = getenv("RG_STEP_KERNEL")) {
        if (!strcmp(force, "group")) h->use_tpe = false;
        else if (!strcmp(force, "tpe") && rg::tpe_supported(*params)) h->use_tpe = true;
    }
    return h;
}

int rg_destroy(rg_handle *h) {
    if (!h) return fail(-1, "handle is NULL");
    delete h;
    return 0;
}

int rg_set_stream(rg_handle *h, void *hip_stream) {
    if (!h) return fail(-1, "handle is NULL");
    h->stream = static_cast<hipStream_t>(hip_stream);
    return 0;
}

int rg_bind_state(rg_handle *h, const rg_state *st) {
    if (!h || !st) return fail(-1, "handle or state is NULL");
    if (!st->poses || !st->carry_dist || !st->episode_steps || !st->reset_count)
        return fail(-20, "poses, carry_dist, episode_steps and reset_count are required");
    switch (h->params.scenario) {
        case RG_SCN_PREDATOR_CAPTURE_PREY:
            if (!st->prey_loc || !st->prey_sensed || !st->prey_captured)
                return fail(-21, "PredatorCapturePrey needs prey_loc, prey_sensed, prey_captured");
            if (reinterpret_cast<uintptr_t>(st->prey_loc) & 7u) return fail(-26, "prey_loc must be 8-byte aligned");
            break;
        case RG_SCN_WAREHOUSE:
            if (!st->loaded) return fail(-21, "Warehouse needs loaded");
            break;
        case RG_SCN_SIMPLE:
            if (!st->prey_loc) return fail(-21, "Simple needs prey_loc (its goal)");
            break;
        case RG_SCN_ARCTIC_TRANSPORT:
            if (!st->grid || !st->goal_col || !st->pixel_type || !st->reached_goal)
                return fail(-21, "ArcticTransport needs grid, goal_col, pixel_type, reached_goal");
            if (reinterpret_cast<uintptr_t>(st->grid) & 3u) return fail(-26, "grid must be 4-byte aligned");
            break;
        default:
            if (!st->load || !st->zone_load || !st->messages)
                return fail(-21, "MaterialTransport needs load, zone_load, messages");
    }
    const int nstat = (st->ep_return != nullptr) + (st->done_return_sum != nullptr) + (st->done_count != nullptr) +
                      (st->done_steps_sum != nullptr);
    if (nstat != 0 && nstat != 4) return fail(-25, "rollout statistics arrays: set all four or none");
    if ((st->next_init != nullptr) != (st->next_episode != nullptr)) return fail(-25, "next_init / next_episode: set both or none");
    if (reinterpret_cast<uintptr_t>(st->next_init) & 15u) return fail(-26, "next_init must be 16-byte aligned");
    h->state = *st;
    h->bound = true;
    // whatever the rebound next_init / next_episode arrays hold was not drawn by this handle under a seed it knows:
    // the first rg_step / rg_rollout after a bind marks every block stale (sync_seed), whatever the caller put there
    h->seed_seen = false;
    return 0;
}

static int fill_args(rg_handle *h, rg::KernelArgs &a) {
    if (!h) return fail(-1, "handle is NULL");
    if (!h->bound) return fail(-22, "rg_bind_state has not been called");
    memset(&a, 0, sizeof(a));
    a.p = h->params;
    a.k = h->consts;
    a.st = h->state;
    a.E = h->num_envs;
    a.num_steps = 1;
    a.env_offset = h->env_offset;
    // the precomputed-reset blocks serve the lane-group kernel (latency regime); the thread-per-env kernel ignores them
    a.next_stride = (!h->use_tpe && h->state.next_init) ? rg::next_init_stride(h->params) : 0;
    return 0;
}

// The blocks of rg_state.next_init are functions of (seed, global env, episode): a new seed makes them stale.
static int sync_seed(rg_handle *h, uint64_t seed) {
    if (h->state.next_episode && (!h->seed_seen || h->last_seed != seed)) {
        // first use after rg_bind_state, or a new seed: every tag <- -1 (one 4 E-byte memset on the stream; the caller's
        // initialisation of next_episode is not trusted)
        const hipError_t err = hipMemsetAsync(h->state.next_episode, 0xFF, sizeof(int32_t) * static_cast<size_t>(h->num_envs), h->stream);
        if (err != hipSuccess) return fail(-30, "hipMemsetAsync(next_episode) failed: %s", hipGetErrorString(err));
        h->seed_seen = true;
        h->last_seed = seed;
    }
    return 0;
}

static int check_io(const rg_step_io *io) {
    if (!io->obs || !io->reward || !io->done || !io->dist_travelled || !io->violation || !io->remaining)
        return fail(-24, "every rg_step_io array except qp_sweeps and the gymma block is required");
    if (reinterpret_cast<uintptr_t>(io->obs) & 15u) return fail(-26, "obs must be 16-byte aligned");
    if (io->elapsed) {
        if (!io->truncated || !io->ended || !io->reward_sum) return fail(-24, "gymma block: elapsed needs truncated, ended and reward_sum");
        if (io->time_limit < 1) return fail(-29, "gymma block: time_limit must be > 0");
    }
    return 0;
}

static int launched(hipError_t err) {
    if (err != hipSuccess) return fail(-30, "kernel launch failed: %s", hipGetErrorString(err));
    return 0;
}

#define RG_ON_DEVICE(h)                                                                           \
    DeviceGuard guard_((h)->device);                                                              \
    if (guard_.err != hipSuccess) return fail(-31, "cannot select the handle's device: %s", hipGetErrorString(guard_.err))

int rg_reset(rg_handle *h, const uint8_t *mask, uint64_t seed, int32_t flags) {
    rg::KernelArgs a;
    if (int rc = fill_args(h, a)) return rc;
    if (flags & ~RG_RESET_BOOK_EPISODE) return fail(-28, "unknown rg_reset flags");
    a.reset_mask = mask;
    a.seed = seed;
    a.reset_flags = flags;
    RG_ON_DEVICE(h);
    return launched(rg::launch_reset(a, h->stream));
}

int rg_step(rg_handle *h, const int32_t *actions, const rg_step_io *io, int32_t auto_reset, uint64_t seed) {
    rg::KernelArgs a;
    if (int rc = fill_args(h, a)) return rc;
    if (!actions || !io) return fail(-23, "actions or io is NULL");
    if (int rc = check_io(io)) return rc;
    a.actions = actions;
    a.io = *io;
    a.auto_reset = auto_reset;
    a.seed = seed;
    RG_ON_DEVICE(h);
    if (int rc = sync_seed(h, seed)) return rc;
    return launched(h->use_tpe ? rg::launch_step_tpe(a, h->stream) : rg::launch_step(a, false, h->stream));
}

int rg_rollout(rg_handle *h, const int32_t *actions, int32_t num_steps, const rg_step_io *io, int32_t auto_reset,
               uint64_t seed) {
    rg::KernelArgs a;
    if (int rc = fill_args(h, a)) return rc;
    if (!actions || !io) return fail(-23, "actions or io is NULL");
    if (num_steps < 1) return fail(-27, "num_steps < 1");
    if (io->elapsed) return fail(-29, "the gymma block of rg_step_io belongs to rg_step (one launch per step)");
    if (int rc = check_io(io)) return rc;
    // (Until round 4 every step's [E][N][D] slice had to keep 16-byte alignment.  Not needed: the only 16-byte observation stores
    // are the 4-float blocks of PredatorCapturePrey without capabilities, whose D is a multiple of 4 -- every slice of that
    // format is aligned whatever E and N are; all other formats are written dword by dword, and the staged copies of the
    // thread-per-env kernel test their destination's alignment themselves (step_tpe.h copy_span / stage_obs_rows).)
    a.actions = actions;
    a.io = *io;
    a.num_steps = num_steps;
    a.auto_reset = auto_reset;
    a.seed = seed;
    RG_ON_DEVICE(h);
    if (int rc = sync_seed(h, seed)) return rc;
    if (!h->use_tpe) return launched(rg::launch_rollout(a, h->stream));
    // thread-per-env: the multi-step kernel holds more values live (313 VGPRs at N = 5: one wave per
    // SIMD); it pays while the batch is at most one wave per SIMD (the latency regime), beyond that
    // num_steps single-step launches are faster (measured at 524288 envs: 166 vs 197 us per step)
    if (h->num_envs <= 65536) return launched(rg::launch_rollout_tpe(a, h->stream));
    a.num_steps = 1;
    for (int32_t k = 0; k < num_steps; ++k) {
        const rg::StepView v = rg::step_view(a, k, h->params.n_agents, h->params.obs_dim);
        rg::KernelArgs ak = a;
        ak.actions = v.actions;
        ak.io = v.io;
        if (int rc = launched(rg::launch_step_tpe(ak, h->stream))) return rc;
    }
    return 0;
}

int rg_step_kernel(const rg_handle *h) {
    if (!h) return fail(-1, "handle is NULL");
    return h->use_tpe ? 1 : 0;
}

int rg_get_obs(rg_handle *h, float *obs) {
    rg::KernelArgs a;
    if (int rc = fill_args(h, a)) return rc;
    if (!obs) return fail(-23, "obs is NULL");
    if (reinterpret_cast<uintptr_t>(obs) & 15u) return fail(-26, "obs must be 16-byte aligned");
    a.io.obs = obs;
    RG_ON_DEVICE(h);
    return launched(rg::launch_step(a, true, h->stream));
}

}  // extern "C"
