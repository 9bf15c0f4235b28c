// kernel_args.h -- the one argument block of every launch (host <-> device contract).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/robogym.h"

namespace rg {

// Scalars derived from the parameter block, computed ONCE on the host (rg_create) in binary32 with
// the same expressions the CPU oracle uses (IEEE +,-,*,/,sqrt: identical on x86 SSE and gfx950;
// this translation unit is compiled with -ffp-contract=off on both sides).
struct Consts {
    float dt, pd, inv_pd, r2, wlim, vmax, wmax, pvl, bml;
    float xmin, xmax, ymin, ymax, coll_off, coll_lim2;
    float thr_pre;         // collision pre-test: threshold on the squared distance of the binary16-rounded points
    float lin_pre;         // the same threshold as a distance (collision limit + rounding bound): the sparse pre-test widens it by travel
    float xc, xh, yc, yh;  // boundary pre-test: arena centre and half extents
};

// Conservative pre-tests of _validate (roboEnv.py:82-94).  The exact float tests run only when a
// pre-test fires; a pre-test may fire needlessly but never misses:
//  * collision: the collision points are rounded toward zero to binary16 pairs.  For |coordinate| <
//    2^(m+1) m the spacing is 2^(m-10) m, so each rounded coordinate is off by less than that, their
//    packed difference by less than twice that plus its own rounding (<= 2^-14 m below 0.25 m), and the
//    distance by sqrt(2) times the per-axis error; `pre_margin` is that bound with 5 % on top.
//  * boundary: a robot moves at most |dt v| (1 + 1e-6) per sub-step, so all pre-update positions of a
//    chunk of C sub-steps lie within (C-1) |dt v| of the first; the test is |x - xc| + margin > xh.
//  * sparse collision pre-test (lane-group kernel): inside a controller period a robot's collision point moves at
//    most m = (|dt v| + coll_off |dt w|)(1 + 1e-5) per sub-step (the body's Euler step, plus the chord of the heading
//    change times the offset), so |f_i(u) - f_j(u)| >= |f_i(u0) - f_j(u0)| - 2 (u - u0) max(m): one test of sub-step u0
//    against lin_pre + 2 span max(m) stands for sub-steps u0 .. u0 + span.  Valid while the widened threshold stays
//    below 0.24 m (the binary16 difference is then below 0.25 m per axis, where its own rounding is <= 2^-14 m).
constexpr float PRE_SLACK = 2e-6f;  // roundings of the position updates and of xc / xh
inline float pre_margin(float max_abs_coordinate) {
    float spacing = 0.0009765625f;  // 2^-10: binary16 spacing in [1, 2)
    for (float lim = 2.0f; lim <= max_abs_coordinate && lim < 1024.0f; lim *= 2.0f) spacing *= 2.0f;
    return 1.05f * 1.41421356f * (2.0f * spacing + 6.103515625e-05f);
}

inline Consts make_consts(const rg_scenario_params &p) {
    Consts k;
    k.dt = p.time_step;
    k.pd = p.projection_distance;
    k.inv_pd = 1.0f / p.projection_distance;
    k.r2 = p.safety_radius * p.safety_radius;
    k.wlim = p.angular_velocity_limit;
    k.vmax = p.max_linear_velocity;
    k.wmax = 2.0f * (p.wheel_radius / p.robot_diameter) * (p.max_linear_velocity / p.wheel_radius);
    k.pvl = p.position_velocity_limit;
    k.bml = p.barrier_magnitude_limit;
    k.xmin = p.bound_x0;
    k.ymin = p.bound_y0;
    k.xmax = p.bound_x0 + p.bound_w;
    k.ymax = p.bound_y0 + p.bound_h;
    const bool off = p.collision_variant == RG_COLLISION_OFFSET;
    k.coll_off = off ? p.collision_offset : 0.0f;
    const float lim = off ? p.collision_diameter : p.robot_diameter;
    k.coll_lim2 = lim * lim;
    const float ax = fmaxf(fabsf(k.xmin), fabsf(k.xmax)), ay = fmaxf(fabsf(k.ymin), fabsf(k.ymax));
    const float lp = lim + pre_margin(fmaxf(ax, ay) + 0.25f);  // robots that left the arena fire the boundary test
    k.thr_pre = lp * lp * 1.00001f;
    k.lin_pre = lp;
    k.xc = 0.5f * (k.xmin + k.xmax);
    k.xh = 0.5f * (k.xmax - k.xmin);
    k.yc = 0.5f * (k.ymin + k.ymax);
    k.yh = 0.5f * (k.ymax - k.ymin);
    return k;
}

struct KernelArgs {
    rg_scenario_params p;
    Consts k;
    rg_state st;
    rg_step_io io;
    const int32_t *actions;
    const uint8_t *reset_mask;
    int32_t E;
    int32_t envs_per_wave; // lane-group kernel: env slots used per wavefront (0 = all 64/GW; fewer for small batches, see launch_step_scn)
    int32_t num_steps;  // env steps per launch (rg_step: 1); io and actions carry a leading dimension of this size
    int32_t auto_reset;
    int32_t reset_flags;  // rg_reset: RG_RESET_*
    int32_t next_stride;  // floats per env of st.next_init (0 = the precomputed-reset blocks are not in use)
    int64_t env_offset;
    uint64_t seed;
};

// floats per env of rg_state.next_init: poses [3N] | prey [2P] / zone loads [2] / terrain [24] + goal column, rounded up to 4
inline int next_init_stride(const rg_scenario_params &p) {
    int w = 3 * p.n_agents;
    if (p.scenario == RG_SCN_PREDATOR_CAPTURE_PREY || p.scenario == RG_SCN_SIMPLE) w += 2 * p.num_prey;
    else if (p.scenario == RG_SCN_MATERIAL_TRANSPORT) w += 2;
    else if (p.scenario == RG_SCN_ARCTIC_TRANSPORT) w += 25;
    return (w + 3) & ~3;
}

hipError_t launch_step(const KernelArgs &a, bool obs_only, hipStream_t stream);
hipError_t launch_reset(const KernelArgs &a, hipStream_t stream);

// the k-th step's slice of the action and output arrays
struct StepView {
    const int32_t *actions;
    rg_step_io io;
};
__host__ __device__ inline StepView step_view(const KernelArgs &a, int k, int n_agents, int obs_dim) {
    StepView v;
    const size_t e = static_cast<size_t>(a.E) * k, en = e * n_agents;
    v.actions = a.actions ? a.actions + en : nullptr;
    v.io.obs = a.io.obs ? a.io.obs + en * obs_dim : nullptr;
    v.io.reward = a.io.reward ? a.io.reward + en : nullptr;
    v.io.done = a.io.done ? a.io.done + e : nullptr;
    v.io.dist_travelled = a.io.dist_travelled ? a.io.dist_travelled + en : nullptr;
    v.io.violation = a.io.violation ? a.io.violation + e : nullptr;
    v.io.remaining = a.io.remaining ? a.io.remaining + e : nullptr;
    v.io.qp_sweeps = a.io.qp_sweeps ? a.io.qp_sweeps + e : nullptr;
    v.io.elapsed = a.io.elapsed;  // state: the same counter for every step of a multi-step launch
    v.io.truncated = a.io.truncated ? a.io.truncated + e : nullptr;
    v.io.ended = a.io.ended ? a.io.ended + e : nullptr;
    v.io.reward_sum = a.io.reward_sum ? a.io.reward_sum + e : nullptr;
    v.io.time_limit = a.io.time_limit;
    v.io.zero_obs_on_end = a.io.zero_obs_on_end;
    return v;
}
// thread-per-env step kernel (robogym_tpe.hip): same results, chosen by the host for large batches
bool tpe_supported(const rg_scenario_params &p);
hipError_t launch_step_tpe(const KernelArgs &a, hipStream_t stream);
// rg_rollout: num_steps env steps per launch (robogym_rollout_group.hip, robogym_rollout_tpe.hip)
hipError_t launch_rollout(const KernelArgs &a, hipStream_t stream);
hipError_t launch_rollout_tpe(const KernelArgs &a, hipStream_t stream);
// the lane-group kernels of the interior-point mode (robogym_kernels_ipm.hip, robogym_rollout_group_ipm.hip); a.envs_per_wave and
// the grid as launch_step_scn computed them
hipError_t launch_step_ipm(const KernelArgs &a, int grid, hipStream_t stream);
hipError_t launch_rollout_ipm(const KernelArgs &a, int grid, hipStream_t stream);

}  // namespace rg
