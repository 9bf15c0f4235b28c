/* ORACLE -- test infrastructure only (see oracle_core.h).  Parameter block of sim_spec_v0. */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>

#define ORC_MAXN 16
#define ORC_MAXP 64
#define ORC_SCN_PCP 0
#define ORC_SCN_WAREHOUSE 1
#define ORC_SCN_MT 2
#define ORC_SCN_SIMPLE 3
#define ORC_SCN_ARCTIC 4

typedef struct orc_params {
    int32_t scenario;
    int32_t n_agents;
    int32_t obs_dim;              /* per-agent observation length D */
    int32_t update_frequency;     /* U: sim sub-iterations per env step */
    int32_t controller_period;    /* 15 (roboEnv.py:63) */
    int32_t max_episode_steps;
    int32_t penalize_violations;
    int32_t barrier_has_unsafe_gain; /* 1: certificate2 ('safe'), 0: certificate ('default') */
    int32_t collision_variant;    /* 0 centre distance <= robot_diameter, 1 offset points <= collision_diameter */
    int32_t capability_aware;
    int32_t num_prey;
    int32_t num_neighbors;
    int32_t torque[ORC_MAXN];
    /* rps constants */
    double time_step, bound_x0, bound_y0, bound_w, bound_h;
    double robot_diameter, wheel_radius, max_linear_velocity;
    double collision_offset, collision_diameter;
    double projection_distance, angular_velocity_limit, position_velocity_limit;
    double barrier_gain, unsafe_barrier_gain, safety_radius, barrier_magnitude_limit;
    double qp_rtol;               /* Hildreth stop: max change <= qp_rtol * max(|u|_inf, magnitude_limit) */
    int32_t qp_max_sweeps;
    int32_t qp_mode;              /* 0: the exact projection (sim_spec_v0).  1: the restated cvxopt interior-point iterate the reference's
                                     stack computes (float64 tier: in cvxopt's operation order; float tier: ipm_spec_v0 = the kernels).
                                     2: ipm_spec_v0 in either precision (float64: the study twin of the spec).  oracle_core.h */
    double ipm_abstol, ipm_reltol, ipm_feastol; /* cvxopt options as rps sets them: abstol 1e-7 (default), reltol = feastol = 1e-2 */
    int32_t ipm_maxiters;                       /* 50 */
    int32_t pad_;
    /* scenario */
    double left, right, up, down;
    double agent_step[ORC_MAXN];      /* step_dist, or MaterialTransport per-agent speed */
    double sensing_radius[ORC_MAXN];
    double capture_radius[ORC_MAXN];
    double time_penalty, sense_reward, capture_reward, violation_reward;
    double load_reward, unload_reward, goal_width;
    double unload_multiplier, load_multiplier, end_goal_width, zone1_radius;
    double reward_scaler;                                      /* Simple */
    double arctic_normal_step, arctic_slow_step, arctic_fast_step; /* ArcticTransport */
    double not_reached_penalty, dist_multiplier;
} orc_params;

/* reset sampler (a17) of sim_spec_v0: grid geometry as the host computes it in float64 from the
 * reference's formulas (misc.py:49-63, warehouse.py:93-98), already rounded to float */
typedef struct orc_grid {
    int32_t nx, ny;
    float spacing, w2, h2, ox1, ox2, oy1, oy2;
} orc_grid;

typedef struct orc_reset_params {
    int32_t scenario, n_agents, num_prey, keep_theta;
    orc_grid agent_grid, prey_grid;
    float zone1_mean, zone1_std, zone2_mean, zone2_std;
} orc_reset_params;

#endif
