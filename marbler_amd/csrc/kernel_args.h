// kernel_args.h -- the one argument block of every launch (host <-> device contract).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/robogym.h"

namespace rg {

// Scalars derived from the parameter block, computed ONCE on the host (rg_create) in binary32 with
// the same expressions the CPU oracle uses (IEEE +,-,*,/,sqrt: identical on x86 SSE and gfx950;
// this translation unit is compiled with -ffp-contract=off on both sides).
struct Consts {
    float dt, pd, inv_pd, r2, wlim, vmax, wmax, pvl, bml;
    float xmin, xmax, ymin, ymax, coll_off, coll_lim2;
    int32_t thr_q;  // collision pre-test threshold on the squared int16 distance (4 LSB margin)
};

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
    // positions are quantised at 4 m <-> 32767 (LSB 0.122 mm); each end point is off by <= 1 LSB per
    // axis, so the distance by < 3 LSB: 4 LSB of margin keep the integer test conservative
    const float lq = lim * 8191.75f + 4.0f;
    k.thr_q = static_cast<int32_t>(lq * lq) + 1;
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
    int32_t auto_reset;
    int64_t env_offset;
    uint64_t seed;
};

hipError_t launch_step(const KernelArgs &a, bool obs_only, hipStream_t stream);
hipError_t launch_reset(const KernelArgs &a, hipStream_t stream);
// thread-per-env step kernel (robogym_tpe.hip): same results, chosen by the host for large batches
bool tpe_supported(const rg_scenario_params &p);
hipError_t launch_step_tpe(const KernelArgs &a, hipStream_t stream);

}  // namespace rg
