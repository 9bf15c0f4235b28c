// robogym_kernels.hip -- the fused env-step kernel for gfx950 (MI355X, CDNA4).
//
// One launch = one env step for E envs: goal generation, U sim sub-iterations (controller with
// the barrier-certificate QP every 15th, collision/boundary validation, Euler integration),
// then the scenario's tracking / observation / reward / termination -- all with the env's
// state in registers.  HBM traffic is the algorithmic I/O only.
//
// Mapping: a lane GROUP of GW lanes (GW = 4, 8 or 16 >= N) owns one env, one lane per agent; a
// 64-lane wavefront carries 64/GW envs; one wavefront per workgroup (no cross-wave sync
// anywhere).  The O(N^2) pair work (collision scan, QP constraint sweeps, neighbour
// distances) runs as GW-1 "rounds": in round k lane a is paired with lane a^k, which is a
// 1-factorisation of the complete graph on the group -- disjoint pairs, so a Gauss-Seidel
// sweep over the QP constraints in this order is pair-parallel yet identical to the
// sequential sweep of the CPU oracle.  Partner data moves by DPP (row-local lane permutes on
// the VALU), never through memory.  Per-env flags reduce with one wave ballot.
//
// No MFMA: there is no dense contraction on this path.  The kernel is a latency-bound
// dependent chain (U x {sincos, 7 pair rounds, sqrt}), so the design goal is the shortest
// per-lane instruction chain, not bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/robogym.h"
#include "kernel_args.h"
#include "sim_math.h"

namespace rg {

// ------------------------------------------------------------------ lane exchange (DPP)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// value held by lane (lane ^ K), K in 1..15, within a 16-lane row
template <int K>
__device__ __forceinline__ float xor_lane(float v) {
    static_assert(K >= 1 && K <= 15, "xor distance");
    constexpr int QP1 = 0xB1, QP2 = 0x4E, QP3 = 0x1B;           // quad_perm [1,0,3,2] [2,3,0,1] [3,2,1,0]
    constexpr int HALF_MIRROR = 0x141, MIRROR = 0x140, ROR8 = 0x128;
    if constexpr (K == 1) return dpp_f<QP1>(v);
    else if constexpr (K == 2) return dpp_f<QP2>(v);
    else if constexpr (K == 3) return dpp_f<QP3>(v);
    else if constexpr (K == 7) return dpp_f<HALF_MIRROR>(v);
    else if constexpr (K == 4) return dpp_f<QP3>(dpp_f<HALF_MIRROR>(v));
    else if constexpr (K == 5) return dpp_f<QP2>(dpp_f<HALF_MIRROR>(v));
    else if constexpr (K == 6) return dpp_f<QP1>(dpp_f<HALF_MIRROR>(v));
    else if constexpr (K == 15) return dpp_f<MIRROR>(v);
    else if constexpr (K == 8) return dpp_f<ROR8>(v);
    else if constexpr (K == 9) return dpp_f<QP1>(dpp_f<ROR8>(v));
    else if constexpr (K == 10) return dpp_f<QP2>(dpp_f<ROR8>(v));
    else if constexpr (K == 11) return dpp_f<QP3>(dpp_f<ROR8>(v));
    else if constexpr (K == 12) return dpp_f<QP3>(dpp_f<MIRROR>(v));
    else if constexpr (K == 13) return dpp_f<QP2>(dpp_f<MIRROR>(v));
    else return dpp_f<QP1>(dpp_f<MIRROR>(v));
}

template <int K>
__device__ __forceinline__ int xor_lane_i(int v) {
    return __builtin_bit_cast(int, xor_lane<K>(__builtin_bit_cast(float, v)));
}

typedef short short2v __attribute__((ext_vector_type(2)));

template <int I, int END, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, END>(f);
    }
}

// max over the lanes of a group (butterfly; every stage is a single DPP permute)
template <int GW>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, xor_lane<1>(v));
    v = fmaxf(v, xor_lane<2>(v));
    if constexpr (GW >= 8) v = fmaxf(v, xor_lane<7>(v));    // quads {0-3} <-> {4-7}: one row_half_mirror
    if constexpr (GW >= 16) v = fmaxf(v, xor_lane<15>(v));  // halves of the row: one row_mirror
    return v;
}

// does any lane of my group have `pred` set?  (one v_cmp -> SGPR pair ballot, then bit tests)
template <int GW>
__device__ __forceinline__ bool group_any(bool pred, int gbase) {
    const unsigned long long m = __ballot(pred);
    constexpr unsigned long long GM = (GW == 64) ? ~0ull : ((1ull << GW) - 1ull);
    return ((m >> gbase) & GM) != 0ull;
}


struct Consts {  // derived scalars, computed in binary32 in the same form as the oracle
    float dt, pd, inv_pd, r2, wlim, vmax, wmax, pvl, bml;
    float xmin, xmax, ymin, ymax, coll_off, coll_lim2;
};
__device__ __forceinline__ Consts make_consts(const rg_scenario_params &p) {
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
    return k;
}

// ------------------------------------------------------------------ controller (a3..a8)
// utilities/controller.py:20-24 over the restated rps closures,
// followed by Robotarium.set_velocities' clipping.  Called in wave-uniform control flow.
template <int GW>
__device__ __forceinline__ int controller(const rg_scenario_params &p, const Consts &k, int N, int ag, bool lane_ok,
                                          bool upd, float x, float y, float c, float s, float gx, float gy, float &v,
                                          float &w) {
    // a4 uni_to_si_states, a5 si_position_controller (gain 1, |dxi| <= 0.15)
    const float xix = x + k.pd * c, xiy = y + k.pd * s;
    float ux = gx - xix, uy = gy - xiy;
    {
        const float nrm = norm2_spec(ux, uy);
        const float sc = k.pvl / nrm;
        const bool clip = nrm > k.pvl;
        ux = clip ? ux * sc : ux;
        uy = clip ? uy * sc : uy;
    }
    // a6 barrier certificate: rows e_ij.(u_j - u_i) <= beta_ij, one per round
    const float bgain = p.barrier_gain, ugain = p.unsafe_barrier_gain, qp_rtol = p.qp_rtol;
    const int qp_cap = p.qp_max_sweeps;
    const bool has_unsafe = p.barrier_has_unsafe_gain != 0;
    float ex[GW - 1], ey[GW - 1], beta[GW - 1], rn2[GW - 1], mu[GW - 1];
    static_for<1, GW>([&](auto KK) {
        constexpr int K = decltype(KK)::value;
        const float pxi = xor_lane<K>(xix), pyi = xor_lane<K>(xiy);
        const float dx = xix - pxi, dy = xiy - pyi;
        const float ee = dx * dx + dy * dy;
        const float h = ee - k.r2;
        const float gain = ((h >= 0.0f) | !has_unsafe) ? bgain : ugain;
        const float b = gain * ((h * h) * h);
        const float n2 = 2.0f * ee;
        const bool ok = lane_ok & ((ag ^ K) < N) & (n2 > 0.0f);
        ex[K - 1] = dx;
        ey[K - 1] = dy;
        beta[K - 1] = 0.5f * b;
        rn2[K - 1] = ok ? 1.0f / n2 : 0.0f;
        mu[K - 1] = 0.0f;
    });
    {   // "Threshold control inputs before QP"
        const float nrm = norm2_spec(ux, uy);
        const float sc = k.bml / nrm;
        const bool clip = nrm > k.bml;
        ux = clip ? ux * sc : ux;
        uy = clip ? uy * sc : uy;
    }
    // Hildreth sweeps; a group drops out when converged, the wave loops while any group is active
    bool active = upd;
    int sweeps = 0, my_sweeps = 0;
    while (__any(active)) {
        float chg = 0.0f;
        static_for<1, GW>([&](auto KK) {
            constexpr int K = decltype(KK)::value;
            // rn2 is 0 for absent pairs and mu starts at 0, so they are no-ops; a converged group is
            // frozen by zeroing its step (d = 0 -> mn = mu -> delta = 0): same arithmetic as the oracle
            const float pux = xor_lane<K>(ux), puy = xor_lane<K>(uy);
            const float r = ex[K - 1] * (pux - ux) + ey[K - 1] * (puy - uy) - beta[K - 1];
            const float d = r * (active ? rn2[K - 1] : 0.0f);
            float mn = mu[K - 1] + d;
            mn = (mn > 0.0f) ? mn : 0.0f;
            const float delta = mn - mu[K - 1];
            mu[K - 1] = mn;
            const float cx = delta * ex[K - 1], cy = delta * ey[K - 1];
            ux = ux + cx;
            uy = uy + cy;
            chg = fmaxf(chg, fmaxf(__builtin_fabsf(cx), __builtin_fabsf(cy)));
        });
        ++sweeps;
        my_sweeps = active ? sweeps : my_sweeps;
        const float um = lane_ok ? fmaxf(__builtin_fabsf(ux), __builtin_fabsf(uy)) : 0.0f;
        const float gchg = group_max<GW>(chg);
        const float gum = fmaxf(k.bml, group_max<GW>(um));
        active = active & (gchg > qp_rtol * gum) & (sweeps < qp_cap);
    }
    // a7 si_to_uni_dyn, a8 set_velocities
    float vv = c * ux + s * uy;
    float ww = k.inv_pd * (-s * ux + c * uy);
    ww = ww > k.wlim ? k.wlim : ww;
    ww = ww < -k.wlim ? -k.wlim : ww;
    vv = vv > k.vmax ? k.vmax : vv;
    vv = vv < -k.vmax ? -k.vmax : vv;
    ww = ww > k.wmax ? k.wmax : ww;
    ww = ww < -k.wmax ? -k.wmax : ww;
    v = upd ? vv : v;
    w = upd ? ww : w;
    return my_sweeps;
}

// ------------------------------------------------------------------ reset sampler (a17)
// misc.py:49-63 -> rps generate_initial_conditions (Appendix A.7): N distinct cells of an
// nx x ny grid, then the scenario's shift.  The reference draws from NumPy's global MT19937;
// here every (global env, episode) pair owns a Philox4x32-10 stream (DESIGN.md "reset").
using GridSpec = rg_grid;

struct Draws {
    uint32_t k0, k1, c0, c1, c2;
    uint32_t blk[4];
    uint32_t cur;   // index of the cached block, 0xFFFFFFFF = none
    uint32_t next;  // next draw index
    __device__ __forceinline__ uint32_t u32() {
        const uint32_t b = next >> 2;
        if (b != cur) {
            philox4x32_10(c0, c1, c2, b, k0, k1, blk);
            cur = b;
        }
        const uint32_t lane = next & 3u;
        ++next;
        return lane == 0 ? blk[0] : lane == 1 ? blk[1] : lane == 2 ? blk[2] : blk[3];
    }
};

// One lane (the group's lane 0) runs the sequential sampler for its env, using `perm` (LDS
// scratch, >= nx*ny bytes) for the partial Fisher-Yates shuffle.
__device__ inline void sample_cells(Draws &d, const GridSpec &g, int count, uint8_t *perm, float *outx, float *outy,
                                    int stride) {
    const int C = g.nx * g.ny;
    for (int i = 0; i < C; ++i) perm[i] = static_cast<uint8_t>(i);
    for (int i = 0; i < count; ++i) {
        const uint32_t r = d.u32();
        const int j = i + static_cast<int>((static_cast<uint64_t>(r) * static_cast<uint32_t>(C - i)) >> 32);
        const uint8_t t = perm[i];
        perm[i] = perm[j];
        perm[j] = t;
        const int cell = perm[i];
        const int cx = cell / g.ny, cy = cell - cx * g.ny;
        const float x = static_cast<float>(cx) * g.spacing - g.w2;
        const float y = static_cast<float>(cy) * g.spacing - g.h2;
        outx[i * stride] = (x + g.ox1) + g.ox2;
        outy[i * stride] = (y + g.oy1) + g.oy2;
    }
}

__device__ __forceinline__ float uniform01(uint32_t r) { return static_cast<float>(r >> 8) * 5.9604644775390625e-08f; }

// int(np.random.normal(mean, std)) by Box-Muller on the spec'd log / sincos
__device__ inline int normal_int(Draws &d, float mean, float stdv) {
    const uint32_t r1 = d.u32(), r2 = d.u32();
    const float u1 = static_cast<float>((r1 >> 8) + 1u) * 5.9604644775390625e-08f;  // (0, 1]
    const float u2 = uniform01(r2);
    const float rad = __builtin_sqrtf(-2.0f * log_spec(u1));
    float sn, cs;
    sincos_spec(u2 * 6.283185482025146484375f - 3.1415927410125732421875f, sn, cs);
    const float z = rad * cs;
    return static_cast<int>(mean + stdv * z);
}


// Resets env e (called by ONE lane per env).  Writes the state arrays in HBM.
template <int SCN>
__device__ inline void reset_env(const KernelArgs &a, int e, uint8_t *perm) {
    const rg_scenario_params &p = a.p;
    const int N = p.n_agents;
    const int32_t episode = a.st.reset_count[e];
    a.st.reset_count[e] = episode + 1;
    const uint64_t ge = static_cast<uint64_t>(a.env_offset + e);
    Draws d;
    d.k0 = static_cast<uint32_t>(a.seed);
    d.k1 = static_cast<uint32_t>(a.seed >> 32);
    d.c0 = static_cast<uint32_t>(ge);
    d.c1 = static_cast<uint32_t>(ge >> 32);
    d.c2 = static_cast<uint32_t>(episode);
    d.cur = 0xFFFFFFFFu;
    d.next = 0;
    float *X = a.st.poses + static_cast<size_t>(e) * 3 * N;
    if constexpr (SCN == RG_SCN_MATERIAL_TRANSPORT) {  // MaterialTransport.py:99-100 come first
        a.st.zone_load[2 * e + 0] = normal_int(d, p.zone1_mean, p.zone1_std);
        a.st.zone_load[2 * e + 1] = normal_int(d, p.zone2_mean, p.zone2_std);
        for (int i = 0; i < 4; ++i) a.st.messages[4 * e + i] = 0;
    }
    sample_cells(d, p.agent_grid, N, perm, X, X + N, 1);
    for (int i = 0; i < N; ++i) {
        const float th = uniform01(d.u32()) * 6.283185482025146484375f - 3.1415927410125732421875f;
        X[2 * N + i] = p.keep_theta ? th : 0.0f;
        a.st.carry_dist[static_cast<size_t>(e) * N + i] = 0.0f;
        if constexpr (SCN == RG_SCN_WAREHOUSE) a.st.loaded[static_cast<size_t>(e) * N + i] = 0;
        if constexpr (SCN == RG_SCN_MATERIAL_TRANSPORT) a.st.load[static_cast<size_t>(e) * N + i] = 0;
    }
    if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) {
        const int P = p.num_prey;
        float *pl = a.st.prey_loc + static_cast<size_t>(e) * P * 2;
        sample_cells(d, p.prey_grid, P, perm, pl, pl + 1, 2);
        for (int i = 0; i < P; ++i) {
            a.st.prey_sensed[static_cast<size_t>(e) * P + i] = 0;
            a.st.prey_captured[static_cast<size_t>(e) * P + i] = 0;
        }
    }
    a.st.episode_steps[e] = 0;
}

// ------------------------------------------------------------------ scenario epilogues
constexpr int WAVE = 64;

struct Lds {  // per-workgroup (= per-wave) scratch
    float prey[WAVE / 4][RG_MAX_PREY * 2];
    uint8_t sensed[WAVE / 4][RG_MAX_PREY];
    uint8_t captured[WAVE / 4][RG_MAX_PREY];
    float ax[WAVE], ay[WAVE];
    int aload[WAVE];
    uint8_t perm[WAVE / 4][64];
};

// K nearest neighbours' own-observations into obs slots 1..K (ascending distance, ties ->
// lower index: the canonical order for misc.py:20-25); K >= N-1: all others in index order.
template <int GW, int OD>
__device__ __forceinline__ void write_neighbour_obs(int N, int Knb, int ag, bool lane_ok, float x, float y,
                                                    const float (&own)[OD], float *obs_row) {
    float d[GW - 1];
    float nb[GW - 1][OD];
    bool ok[GW - 1];
    int rank[GW - 1];
    static_for<1, GW>([&](auto KK) {
        constexpr int K = decltype(KK)::value;
        const float px = xor_lane<K>(x), py = xor_lane<K>(y);
        const float dx = px - x, dy = py - y;
        d[K - 1] = norm2_spec(dx, dy);
        ok[K - 1] = lane_ok & ((ag ^ K) < N);
        rank[K - 1] = 0;
#pragma unroll
        for (int c = 0; c < OD; ++c) nb[K - 1][c] = xor_lane<K>(own[c]);
    });
    const bool all_others = Knb >= N - 1;
    // rank of partner K among the valid partners: one comparison per unordered pair (Q < K)
    static_for<2, GW>([&](auto KK) {
        constexpr int K = decltype(KK)::value;
        static_for<1, K>([&](auto QQ) {
            constexpr int Q = decltype(QQ)::value;
            // is Q ahead of K?  distance first, then the lower agent index
            const bool q_first = (d[Q - 1] < d[K - 1]) | ((d[Q - 1] == d[K - 1]) & ((ag ^ Q) < (ag ^ K)));
            const bool both = ok[Q - 1] & ok[K - 1];
            rank[K - 1] += (both & q_first) ? 1 : 0;
            rank[Q - 1] += (both & !q_first) ? 1 : 0;
        });
    });
    static_for<1, GW>([&](auto KK) {
        constexpr int K = decltype(KK)::value;
        const int j = ag ^ K;
        const int slot = all_others ? (j < ag ? j : j - 1) : rank[K - 1];
        if (ok[K - 1] & (all_others | (slot < Knb))) {
            float *o = obs_row + (slot + 1) * OD;
#pragma unroll
            for (int c = 0; c < OD; ++c) o[c] = nb[K - 1][c];
        }
    });
}

// ------------------------------------------------------------------ the step kernel
template <int SCN, int GW, bool OBS_ONLY>
__global__ __launch_bounds__(WAVE) void step_kernel(const KernelArgs a) {
    constexpr int EPW = WAVE / GW;  // envs per wave
    __shared__ Lds lds;
    const rg_scenario_params &p = a.p;
    const Consts k = make_consts(p);
    const int N = p.n_agents;
    const int lane = threadIdx.x;
    const int ag = lane & (GW - 1);
    const int g = lane / GW;
    const int gbase = lane & ~(GW - 1);
    const int e = blockIdx.x * EPW + g;
    const bool env_ok = e < a.E;
    const bool lane_ok = env_ok && ag < N;
    const size_t eN = static_cast<size_t>(e) * N;

    // ---- loads (coalesced: a wave covers EPW consecutive envs = one contiguous span per array)
    float x = 0.0f, y = 0.0f, th = 0.0f, carry = 0.0f;
    int act = 4;
    if (lane_ok) {
        const float *X = a.st.poses + eN * 3;
        x = X[ag];
        y = X[N + ag];
        th = X[2 * N + ag];
        if constexpr (!OBS_ONLY) {
            carry = a.st.carry_dist[eN + ag];
            act = a.actions[eN + ag];
        }
    }
    if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) {  // stage the env's prey block in LDS
        const int P = p.num_prey;
        if (env_ok) {
            for (int i = ag; i < 2 * P; i += GW) lds.prey[g][i] = a.st.prey_loc[static_cast<size_t>(e) * 2 * P + i];
            for (int i = ag; i < P; i += GW) {
                lds.sensed[g][i] = a.st.prey_sensed[static_cast<size_t>(e) * P + i];
                lds.captured[g][i] = a.st.prey_captured[static_cast<size_t>(e) * P + i];
            }
        }
    }

    int viol = 0, max_sweeps = 0;
    float dist = 0.0f;
    if constexpr (!OBS_ONLY) {
        // ---- a1 goal generation (agent.py:48-76, warehouse.py:19-45, MaterialTransport.py:19-46)
        float gx = x, gy = y;
        {
            const int mv = (SCN == RG_SCN_MATERIAL_TRANSPORT) ? act / 4 : act;
            const float sd = p.agent_step[ag];
            if (mv == 0) {
                const float t = gx - sd;
                gx = t > p.left ? t : p.left;
                gy = clamp_spec(gy, p.up, p.down);
            } else if (mv == 1) {
                const float t = gx + sd;
                gx = t < p.right ? t : p.right;
                gy = clamp_spec(gy, p.up, p.down);
            } else if (mv == 2) {
                gx = clamp_spec(gx, p.left, p.right);
                const float t = gy - sd;
                gy = t > p.up ? t : p.up;
            } else if (mv == 3) {
                gx = clamp_spec(gx, p.left, p.right);
                const float t = gy + sd;
                gy = t < p.down ? t : p.down;
            } else {
                gx = clamp_spec(gx, p.left, p.right);
                gy = clamp_spec(gy, p.up, p.down);
            }
        }
        // ---- a2 roboEnv.step (utilities/roboEnv.py:52-94), float spec of oracle/oracle_core.h, one
        // CONTROLLER PERIOD (<= 15 sub-steps with v, w held) at a time: theta and dist_travelled
        // advance once per period by fma; inside the period only x, y and (cos, sin) move.
        //
        // _validate every sub-step: the exact test (7 DPP rounds of float math) runs only in a rare
        // wave-uniform branch.  The common path is a conservative integer pre-test on positions
        // quantised to int16 pairs (4 m <-> 32767, LSB 0.12 mm): one DPP + v_pk_sub_i16 + v_dot2 per
        // pair round, with a 4 LSB margin on the distance so it can never miss a collision the float
        // test would flag.  Absent lanes / finished envs sit on far-apart ghost points.
        const float lim = __builtin_sqrtf(k.coll_lim2);
        const float lq = __builtin_fmaf(lim, 8191.75f, 4.0f);
        const int thr_q = static_cast<int>(lq * lq) + 1;
        const int ghost_q = (32767 & 0xFFFF) | (((-28000 + 3500 * ag) & 0xFFFF) << 16);  // >= 0.43 m apart, > 2 m from the arena
        float v = 0.0f, w = 0.0f, s = 0.0f, c = 1.0f;
        float acc = carry, last = 0.0f;  // dist incl. the pending sub-step; length of the last sub-step
        bool dead = false;               // group-uniform: the env hit a violation (roboEnv.py:92-94)
        float fin_x = 0.0f, fin_y = 0.0f;
        const bool penalize = p.penalize_violations != 0;
        const int U = p.update_frequency, period = p.controller_period;
        for (int it0 = 0; it0 < U; it0 += period) {
            const int n = (U - it0) < period ? (U - it0) : period;
            sincos_spec(th, s, c);
            const int sw = controller<GW>(p, k, N, ag, lane_ok, env_ok & !dead, x, y, c, s, gx, gy, v, w);
            max_sweeps = sw > max_sweeps ? sw : max_sweeps;
            const float dtv = k.dt * v, dtw = k.dt * w;
            float sd, cd;
            sincos_spec(dtw, sd, cd);
            int n_exec = n;           // sub-steps this env executes in this period
            bool died_now = false;
            for (int j = 0; j < n; ++j) {
                // a10 _validate on the pre-update poses
                const bool bnd = lane_ok & !dead & ((x < k.xmin) | (x > k.xmax) | (y < k.ymin) | (y > k.ymax));
                const float fx = __builtin_fmaf(k.coll_off, c, x), fy = __builtin_fmaf(k.coll_off, s, y);
                const int q_real = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pknorm_i16(fx * 0.25f, fy * 0.25f));
                const int q = (lane_ok & !dead) ? q_real : ghost_q;
                int dmin = 0x7FFFFFFF;
                static_for<1, GW>([&](auto KK) {
                    constexpr int K = decltype(KK)::value;
                    const short2v dq = __builtin_elementwise_sub_sat(__builtin_bit_cast(short2v, q),
                                                                     __builtin_bit_cast(short2v, xor_lane_i<K>(q)));
                    const int d2 = __builtin_amdgcn_sdot2(dq, dq, 0, false);
                    dmin = d2 < dmin ? d2 : dmin;
                });
                const bool near = dmin <= thr_q;
                if (penalize && __any(near | bnd)) {  // rare: exact float test (roboEnv.py:82-94)
                    bool col = false;
                    static_for<1, GW>([&](auto KK) {
                        constexpr int K = decltype(KK)::value;
                        const float dx = fx - xor_lane<K>(fx), dy = fy - xor_lane<K>(fy);
                        col = col | (((ag ^ K) < N) & (dx * dx + dy * dy <= k.coll_lim2));
                    });
                    col = col & lane_ok & !dead;
                    const int code = (group_any<GW>(col, gbase) ? 1 : 0) | (group_any<GW>(bnd, gbase) ? 2 : 0);
                    if (env_ok & !dead & (code != 0)) {
                        viol = code;
                        n_exec = j + 1;
                        died_now = true;
                        dead = true;
                        fin_x = __builtin_fmaf(c, dtv, x);  // this sub-step is still integrated
                        fin_y = __builtin_fmaf(s, dtv, y);
                    }
                }
                // Euler step (Appendix A.4); rotate (cos, sin) by dt*w
                x = __builtin_fmaf(c, dtv, x);
                y = __builtin_fmaf(s, dtv, y);
                const float cn = __builtin_fmaf(c, cd, -(s * sd));
                const float sn = __builtin_fmaf(s, cd, c * sd);
                c = cn;
                s = sn;
            }
            // period end: heading and distance for the sub-steps this env executed
            const bool upd = env_ok & (!dead | died_now);
            const float ne = static_cast<float>(n_exec);
            const float adv = __builtin_fabsf(dtv);
            th = upd ? wrap_spec(__builtin_fmaf(ne, dtw, th)) : th;
            acc = upd ? __builtin_fmaf(ne, adv, acc) : acc;
            last = upd ? adv : last;
            if (!__any(env_ok & !dead)) break;
        }
        if (dead) {
            x = fin_x;
            y = fin_y;
        }
        dist = viol ? acc : acc - last;
        carry = last;
    }

    __syncthreads();  // LDS prey block visible (single-wave workgroup: compiles to waitcnt + s_barrier)

    // ---- scenario epilogue
    const int D = p.obs_dim;
    float *obs_row = a.io.obs + (eN + ag) * D;
    int steps = 0;
    if (env_ok) steps = a.st.episode_steps[e] + (OBS_ONLY ? 0 : 1);
    bool done = false;
    int remaining = -1;
    float reward = 0.0f;

    if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) {
        const int P = p.num_prey;
        const float sr = p.sensing_radius[ag], cr = p.capture_radius[ag];
        int unseen0 = 0, left0 = 0, unseen1 = 0, left1 = 0;
        float closest = -1.0f, qx = -5.0f, qy = -5.0f;
        for (int i = 0; i < P; ++i) {
            const float plx = lds.prey[g][2 * i], ply = lds.prey[g][2 * i + 1];
            const float d = norm2_spec(x - plx, y - ply);
            bool sen = lds.sensed[g][i] != 0, cap = lds.captured[g][i] != 0;
            unseen0 += !sen;
            left0 += !cap;
            if constexpr (!OBS_ONLY) {  // a11 _update_tracking_and_locations
                const bool any_s = group_any<GW>(lane_ok && d <= sr, gbase);
                const bool any_c = group_any<GW>(lane_ok && act == 4 && d <= cr, gbase);
                if (!cap) {
                    if (!sen && any_s) sen = true;
                    if (sen && any_c) cap = true;
                }
                if (lane_ok && ag == 0) {
                    a.st.prey_sensed[static_cast<size_t>(e) * P + i] = sen;
                    a.st.prey_captured[static_cast<size_t>(e) * P + i] = cap;
                }
            }
            unseen1 += !sen;
            left1 += !cap;
            // a13 own observation: nearest uncaptured prey within the agent's own sensing radius
            if (!cap && d <= sr && (d < closest || closest == -1.0f)) {
                qx = plx;
                qy = ply;
                closest = d;
            }
        }
        if (p.capability_aware) {
            const float own[6] = {x, y, qx, qy, sr, cr};
            if (lane_ok) {
#pragma unroll
                for (int c = 0; c < 6; ++c) obs_row[c] = own[c];
            }
            write_neighbour_obs<GW, 6>(N, p.num_neighbors, ag, lane_ok, x, y, own, obs_row);
        } else {
            const float own[4] = {x, y, qx, qy};
            if (lane_ok) *reinterpret_cast<float4 *>(obs_row) = make_float4(x, y, qx, qy);
            write_neighbour_obs<GW, 4>(N, p.num_neighbors, ag, lane_ok, x, y, own, obs_row);
        }
        if constexpr (!OBS_ONLY) {  // a14 reward / termination
            if (viol) {
                reward = p.violation_reward;
                done = true;
            } else {
                reward = 0.0f;
                reward = reward + static_cast<float>(unseen0 - unseen1) * p.sense_reward;
                reward = reward + static_cast<float>(left0 - left1) * p.capture_reward;
                reward = reward + p.time_penalty;
                if (steps > p.max_episode_steps || left1 == 0) {
                    done = true;
                    remaining = left1;
                }
            }
        }
    } else if constexpr (SCN == RG_SCN_WAREHOUSE) {  // a15
        uint8_t loaded = 0;
        if (lane_ok) loaded = a.st.loaded[eN + ag];
        const float own[3] = {x, y, loaded ? 1.0f : 0.0f};
        if (lane_ok) {
            obs_row[0] = own[0];
            obs_row[1] = own[1];
            obs_row[2] = own[2];
        }
        write_neighbour_obs<GW, 3>(N, p.num_neighbors, ag, lane_ok, x, y, own, obs_row);
        if constexpr (!OBS_ONLY) {
            if (viol) {
                reward = p.violation_reward;
                done = true;
            } else {
                const bool green = (ag % 2) == 0;
                if (loaded) {
                    if (x < -1.5f + p.goal_width && ((green && y > 0.0f) || (!green && y <= 0.0f))) {
                        reward = p.unload_reward;
                        loaded = 0;
                    }
                } else {
                    if (x > 1.5f - p.goal_width && ((!green && y > 0.0f) || (green && y <= 0.0f))) {
                        reward = p.load_reward;
                        loaded = 1;
                    }
                }
                done = steps > p.max_episode_steps;
                if (lane_ok) a.st.loaded[eN + ag] = loaded;
            }
        }
    } else {  // a16 MaterialTransport
        int msg[4] = {0, 0, 0, 0};
        int zone0 = 0, zone1 = 0, load = 0;
        if (env_ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                msg[i] = a.st.messages[4 * e + i];
                if constexpr (!OBS_ONLY)
                    if (i < N) msg[i] = a.actions[eN + i] % 4;
            }
            zone0 = a.st.zone_load[2 * e];
            zone1 = a.st.zone_load[2 * e + 1];
        }
        if (lane_ok) load = a.st.load[eN + ag];
        if (lane_ok) {
            obs_row[0] = x;
            obs_row[1] = y;
            obs_row[2] = static_cast<float>(load);
            obs_row[3] = static_cast<float>(zone0);
            obs_row[4] = static_cast<float>(zone1);
#pragma unroll
            for (int i = 0; i < 4; ++i) obs_row[5 + i] = static_cast<float>(msg[i]);
            if (p.capability_aware) {
                obs_row[9] = static_cast<float>(p.torque[ag]);
                obs_row[10] = p.agent_step[ag];
            }
        }
        if constexpr (!OBS_ONLY) {
            lds.ax[lane] = x;
            lds.ay[lane] = y;
            lds.aload[lane] = load;
            __syncthreads();
            if (viol) {
                reward = p.violation_reward;
                done = true;
            } else {
                // zone depletion is order-dependent across agents (MaterialTransport.py:161-189):
                // every lane replays the env's sequential loop from the LDS copy
                reward = p.time_penalty;
                const float egw = p.end_goal_width;
                bool any_load = false;
                for (int j = 0; j < N; ++j) {
                    const float jx = lds.ax[gbase + j], jy = lds.ay[gbase + j];
                    int jl = lds.aload[gbase + j];
                    const int tq = p.torque[j];
                    if (jl > 0) {
                        if (jx < -1.5f + egw) {
                            reward = reward + static_cast<float>(jl) * p.unload_multiplier;
                            jl = 0;
                        }
                    } else {
                        if (jx > 1.5f - egw) {
                            if (zone1 > tq) {
                                jl = tq;
                                zone1 -= tq;
                            } else {
                                jl = zone1;
                                zone1 = 0;
                            }
                            reward = reward + static_cast<float>(jl) * p.load_multiplier;
                        } else if (norm2_spec(jx, jy) <= p.zone1_radius) {
                            if (zone0 > tq) {
                                jl = tq;
                                zone0 -= tq;
                            } else {
                                jl = zone0;
                                zone0 = 0;
                            }
                            reward = reward + static_cast<float>(jl) * p.load_multiplier;
                        }
                    }
                    if (j == ag) load = jl;
                    any_load = any_load || (jl != 0);
                }
                done = steps > p.max_episode_steps;
                if (!done) done = (zone0 == 0 && zone1 == 0 && !any_load);
            }
            int total = 0;
            {   // info['remaining'] = zone loads + agent loads (after the update)
                lds.aload[lane] = lane_ok ? load : 0;
                __syncthreads();
                for (int j = 0; j < N; ++j) total += lds.aload[gbase + j];
            }
            if (done) remaining = zone0 + zone1 + total;
            if (lane_ok) a.st.load[eN + ag] = load;
            if (lane_ok && ag == 0) {
                a.st.zone_load[2 * e] = zone0;
                a.st.zone_load[2 * e + 1] = zone1;
            }
            if (lane_ok && ag < 4) a.st.messages[4 * e + ag] = msg[ag == 0 ? 0 : ag == 1 ? 1 : ag == 2 ? 2 : 3];
        }
    }

    if constexpr (!OBS_ONLY) {
        // sum of the agents' rewards in agent order (only read when shared_reward == 0)
        float rsum = 0.0f;
        if (a.st.ep_return && !p.shared_reward) {
            lds.ax[lane] = lane_ok ? reward : 0.0f;
            __syncthreads();
            for (int j = 0; j < N; ++j) rsum = rsum + lds.ax[gbase + j];
        }
        // ---- stores
        if (lane_ok) {
            float *X = a.st.poses + eN * 3;
            X[ag] = x;
            X[N + ag] = y;
            X[2 * N + ag] = th;
            a.st.carry_dist[eN + ag] = carry;
            a.io.reward[eN + ag] = reward;
            a.io.dist_travelled[eN + ag] = dist;
            if (ag == 0) {
                a.st.episode_steps[e] = steps;
                if (a.st.ep_return) {  // misc.py:178-185: episodeReward += reward[0] | sum(reward)
                    float ret = a.st.ep_return[e] + (p.shared_reward ? reward : rsum);
                    if (done) {
                        a.st.done_return_sum[e] = a.st.done_return_sum[e] + ret;
                        a.st.done_count[e] = a.st.done_count[e] + 1;
                        a.st.done_steps_sum[e] = a.st.done_steps_sum[e] + steps;
                        ret = 0.0f;
                    }
                    a.st.ep_return[e] = ret;
                }
                a.io.done[e] = done ? 1 : 0;
                a.io.violation[e] = static_cast<uint8_t>(viol);
                a.io.remaining[e] = remaining;
                if (a.io.qp_sweeps) a.io.qp_sweeps[e] = max_sweeps;
            }
        }
        // ---- fused auto-reset of finished envs (scenario.reset(), not hot: ~1/80 steps)
        if (a.auto_reset) {
            __syncthreads();  // this wave's state stores are issued; the resetting lane rewrites them
            if (lane_ok && ag == 0 && done) reset_env<SCN>(a, e, lds.perm[g]);
        }
    }
}

template <int SCN>
__global__ __launch_bounds__(WAVE) void reset_kernel(const KernelArgs a) {
    __shared__ uint8_t perm[WAVE][64];
    const int e = blockIdx.x * WAVE + threadIdx.x;
    if (e >= a.E) return;
    if (a.reset_mask && !a.reset_mask[e]) return;
    reset_env<SCN>(a, e, perm[threadIdx.x]);
}

}  // namespace rg

// ------------------------------------------------------------------ host side: launch dispatch
namespace rg {

template <int SCN, bool OBS_ONLY>
static hipError_t launch_step_scn(const KernelArgs &a, hipStream_t stream) {
    const int N = a.p.n_agents;
    if (N <= 4) {
        constexpr int GW = 4;
        const int grid = (a.E + WAVE / GW - 1) / (WAVE / GW);
        hipLaunchKernelGGL((step_kernel<SCN, GW, OBS_ONLY>), dim3(grid), dim3(WAVE), 0, stream, a);
    } else if (N <= 8) {
        constexpr int GW = 8;
        const int grid = (a.E + WAVE / GW - 1) / (WAVE / GW);
        hipLaunchKernelGGL((step_kernel<SCN, GW, OBS_ONLY>), dim3(grid), dim3(WAVE), 0, stream, a);
    } else {
        constexpr int GW = 16;
        const int grid = (a.E + WAVE / GW - 1) / (WAVE / GW);
        hipLaunchKernelGGL((step_kernel<SCN, GW, OBS_ONLY>), dim3(grid), dim3(WAVE), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_step(const KernelArgs &a, bool obs_only, hipStream_t stream) {
    switch (a.p.scenario) {
        case RG_SCN_PREDATOR_CAPTURE_PREY:
            return obs_only ? launch_step_scn<RG_SCN_PREDATOR_CAPTURE_PREY, true>(a, stream)
                            : launch_step_scn<RG_SCN_PREDATOR_CAPTURE_PREY, false>(a, stream);
        case RG_SCN_WAREHOUSE:
            return obs_only ? launch_step_scn<RG_SCN_WAREHOUSE, true>(a, stream)
                            : launch_step_scn<RG_SCN_WAREHOUSE, false>(a, stream);
        case RG_SCN_MATERIAL_TRANSPORT:
            return obs_only ? launch_step_scn<RG_SCN_MATERIAL_TRANSPORT, true>(a, stream)
                            : launch_step_scn<RG_SCN_MATERIAL_TRANSPORT, false>(a, stream);
        default:
            return hipErrorInvalidValue;
    }
}

hipError_t launch_reset(const KernelArgs &a, hipStream_t stream) {
    const int grid = (a.E + WAVE - 1) / WAVE;
    switch (a.p.scenario) {
        case RG_SCN_PREDATOR_CAPTURE_PREY:
            hipLaunchKernelGGL((reset_kernel<RG_SCN_PREDATOR_CAPTURE_PREY>), dim3(grid), dim3(WAVE), 0, stream, a);
            break;
        case RG_SCN_WAREHOUSE:
            hipLaunchKernelGGL((reset_kernel<RG_SCN_WAREHOUSE>), dim3(grid), dim3(WAVE), 0, stream, a);
            break;
        case RG_SCN_MATERIAL_TRANSPORT:
            hipLaunchKernelGGL((reset_kernel<RG_SCN_MATERIAL_TRANSPORT>), dim3(grid), dim3(WAVE), 0, stream, a);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rg
