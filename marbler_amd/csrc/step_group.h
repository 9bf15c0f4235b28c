// step_group.h -- the fused env-step kernel for gfx950 (MI355X, CDNA4), lane-group mapping.
// (Templates only; instantiated by robogym_kernels.hip for rg_step / rg_get_obs / rg_reset and by
// robogym_rollout_group.hip for rg_rollout.)
//
// One launch = one env step for E envs: goal generation, U sim sub-iterations (controller with
// the barrier-certificate QP every 15th, collision/boundary validation, Euler integration),
// then the scenario's tracking / observation / reward / termination, and the reset of envs that
// finished -- all with the env's state in registers.  HBM traffic is the algorithmic I/O only
// (measured with rocprofv3 FETCH_SIZE / WRITE_SIZE in profiles/).
//
// Mapping: a lane GROUP of GW lanes (GW = 4, 8 or 16 >= N) owns one env, one lane per agent; a
// 64-lane wavefront carries 64/GW envs; one wavefront per workgroup (no cross-wave sync
// anywhere).  The O(N^2) pair work (collision scan, QP constraint sweeps) runs as GW-1
// "rounds": in round k lane a is paired with lane a^k, a 1-factorisation of the complete graph
// on the group -- disjoint pairs, so a Gauss-Seidel sweep over the QP constraints in this
// order is pair-parallel yet identical to the sequential sweep of the CPU oracle.  Partner
// data moves by DPP (row-local lane permutes on the VALU), never through memory; per-env
// reductions are DPP butterflies or one wave ballot; the per-env prey block and the agents'
// own-observation rows are staged in LDS.
//
// No MFMA: there is no dense contraction on this path.  At the benchmark size (4096 envs =
// 512 wavefronts on 1024 SIMDs) the kernel is a latency-bound dependent chain per wavefront, so
// the design goal is the shortest per-lane instruction chain with independent work interleaved
// (sub-steps are processed in chunks of 5 for ILP), not bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "device_common.h"
#include "ipm_qp.h"
#include "probes/diag.h"   // diagnostic hooks: every RG_* macro below expands to nothing in the shipped build

namespace rg {

// Output stores (observation rows, rewards, distances): plain stores (non-temporal ones were tried: NOTEBOOK.md section 4.4)
__device__ __forceinline__ void out_store4(float *p, float a, float b, float c, float d) {
    *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void out_store1(float *p, float a) {
    *p = a;
}

// sub-steps validated together (ILP across independent test chains).  The controller periods of the
// reference's configurations are 15 and 14 sub-steps: three chunks of 5, or two and a remainder chunk of 4.
constexpr int CHUNK = RG_CHUNK;
static_assert(CHUNK >= 1 && CHUNK <= 5, "the sparse collision pre-test covers sub-steps 0..2 and 3..4 of a chunk: longer chunks would go untested");

// ------------------------------------------------------------------ controller (a3..a8)
// utilities/controller.py:20-24 over the restated rps closures,
// followed by Robotarium.set_velocities' clipping.  Called in wave-uniform control flow.
// QPM: how the certificate's QP is evaluated (include/robogym.h RG_QP_*): 0 the exact projection by the Hildreth sweeps below,
// 1 cvxopt's interior-point iterate (ipm_qp.h; qp = the wave's LDS of that mode: one record per lane, one workspace per env).
template <int GW, int QPM = 0, typename QpLds = void>
__device__ __forceinline__ int controller(const rg_scenario_params &p, const Consts &k, int N, int ag, bool lane_ok,
                                          bool upd, float x, float y, float c, float s, float gx, float gy, float &v,
                                          float &w, QpLds *qp RG_CTRL_TICKS_PARAM) {
    RG_CTRL_BEGIN(x, y, c, s)
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
    if constexpr (QPM == RG_QP_CVXOPT) {
        // a6 as the reference's stack evaluates it: "threshold control inputs before QP" (decided on squares, as below), then the
        // env's records gathered in LDS and the interior-point iteration of ipm_qp.h, the same in every lane of the group
        {
            const float n2u = ux * ux + uy * uy;
            const bool clip = n2u > k.bml * k.bml;
            if (__any(clip)) {
                const float sc = k.bml / __builtin_sqrtf(n2u);
                ux = clip ? ux * sc : ux;
                uy = clip ? uy * sc : uy;
            }
        }
        const int lane_ = threadIdx.x;
        qp->rec[lane_] = make_float4(xix, xiy, ux, uy);
        __syncthreads();   // (one wave per workgroup: the records are visible)
        int iters = 0;
        if (upd) iters = ipm::solve_qp_n<GW>(N, ipm::make_consts(p), qp->rec + (lane_ & ~(GW - 1)), qp->ws[lane_ / GW], ag);
        __syncthreads();
        const float4 r = qp->rec[lane_];
        ux = r.z;
        uy = r.w;
        RG_CTRL_TICK(1);
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
        RG_CTRL_TICK(2);
        return iters;
    }
    // a6 barrier certificate: rows e_ij.(u_j - u_i) <= beta_ij, one per round; the exact projection
    // by Hildreth sweeps with vector-extrapolation restarts (algorithm and derivation: oracle/oracle_core.h
    // barrier_qp).  Per round: f = e/n2, bp = beta/n2, emax = max(|ex|, |ey|); absent pairs have
    // f = bp = emax = 0 and mu = 0, which makes every update of theirs an exact no-op.
    const float bgain = p.barrier_gain, ugain = p.unsafe_barrier_gain, qp_rtol = p.qp_rtol;
    const int qp_cap = p.qp_max_sweeps;
    const bool has_unsafe = p.barrier_has_unsafe_gain != 0;
    float ex[GW - 1], ey[GW - 1], fx[GW - 1], fy[GW - 1], bp[GW - 1], emax[GW - 1];
    float mu[GW - 1], dmu[GW - 1];  // dmu: the change of mu over sweep 2 (d1), then over sweep 3 (d2), of a block of four sweeps
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
        const float rn2 = ok ? 1.0f / n2 : 0.0f;
        ex[K - 1] = dx;
        ey[K - 1] = dy;
        fx[K - 1] = dx * rn2;
        fy[K - 1] = dy * rn2;
        bp[K - 1] = (0.5f * b) * rn2;
        emax[K - 1] = ok ? fmaxf(__builtin_fabsf(dx), __builtin_fabsf(dy)) : 0.0f;
        mu[K - 1] = dmu[K - 1] = 0.0f;
    });
    RG_CTRL_PIN_ROUNDS(GW, fx, bp, emax)
    RG_CTRL_TICK(0);  // position controller + pair constants
    {   // "Threshold control inputs before QP": decided on squares; never taken after the 0.15 clip
        const float n2u = ux * ux + uy * uy;
        const bool clip = n2u > k.bml * k.bml;
        if (__any(clip)) {
            const float sc = k.bml / __builtin_sqrtf(n2u);
            ux = clip ? ux * sc : ux;
            uy = clip ? uy * sc : uy;
        }
    }
    const float uhx = ux, uhy = uy;
    // A group drops out when converged (its lanes are exec-masked for the whole sweep body: groups
    // are uniform, so an active lane never reads a masked partner); the wave loops while any group
    // is active.
    bool active = upd;
    int sweeps = 0, my_sweeps = 0;
    // one sweep over the GW-1 rounds + the convergence test; PHASE = sweep number mod 4 selects the
    // restart bookkeeping: d1 and d2 of oracle_core.h (the changes of the multipliers over sweeps 2 and 3 of a block of four)
    // are these sweeps' own deltas -- nothing touches mu between them -- so sweep 2 records its delta, sweep 3 forms the two
    // inner products <d2 - d1, d2>, <d2 - d1, d2 - d1> while it runs (independent of its dependent chain: free issue slots)
    // and leaves d2 in place of d1 for the restart
    auto sweep = [&](auto PH) {
        constexpr int PHASE = decltype(PH)::value;
        ++sweeps;
        if (active) {
            float chg = 0.0f, pa = 0.0f, pb = 0.0f;
            static_for<1, GW>([&](auto KK) {
                constexpr int K = decltype(KK)::value;
                const float pux = xor_lane<K>(ux), puy = xor_lane<K>(uy);
                const float c0 = mu[K - 1] - bp[K - 1];
                const float t = __builtin_fmaf(fy[K - 1], puy - uy, c0);
                float mn = __builtin_fmaf(fx[K - 1], pux - ux, t);
                mn = (mn > 0.0f) ? mn : 0.0f;
                const float delta = mn - mu[K - 1];
                mu[K - 1] = mn;
                ux = __builtin_fmaf(delta, ex[K - 1], ux);
                uy = __builtin_fmaf(delta, ey[K - 1], uy);
                chg = fmaxf(chg, __builtin_fabsf(delta) * emax[K - 1]);
                if constexpr (PHASE == 2) dmu[K - 1] = delta;
                if constexpr (PHASE == 3) {
                    const float dd = delta - dmu[K - 1];
                    pa = __builtin_fmaf(dd, delta, pa);
                    pb = __builtin_fmaf(dd, dd, pb);
                    dmu[K - 1] = delta;
                }
            });
            my_sweeps = sweeps;
            const float um = lane_ok ? fmaxf(__builtin_fabsf(ux), __builtin_fabsf(uy)) : 0.0f;
            const float gchg = group_max_nonneg<GW>(chg);
            const float gum = fmaxf(k.bml, group_max_nonneg<GW>(um));
            active = (gchg > qp_rtol * gum) & (sweeps < qp_cap);
            if constexpr (PHASE == 3) {
                if (active) {  // restart: the multipliers extrapolated along their last change, u rebuilt from them
                    const float ga = group_sum<GW>(pa), gb = group_sum<GW>(pb);
                    const bool ok = (gb > 0.0f) & (ga < 0.0f) & (-ga < 32.0f * gb);
                    const float gam = ok ? ga / gb : 0.0f;  // the one division of a restart
                    float sx = uhx, sy = uhy;
                    static_for<1, GW>([&](auto KK) {
                        constexpr int K = decltype(KK)::value;
                        float m = __builtin_fmaf(-gam, dmu[K - 1], mu[K - 1]);
                        m = (m > 0.0f) ? m : 0.0f;
                        mu[K - 1] = m;
                        sx = __builtin_fmaf(m, ex[K - 1], sx);
                        sy = __builtin_fmaf(m, ey[K - 1], sy);
                    });
                    ux = sx;
                    uy = sy;
                }
            }
        }
    };
    while (__any(active)) {
        sweep(std::integral_constant<int, 1>{});
        if (!__any(active)) break;
        sweep(std::integral_constant<int, 2>{});
        if (!__any(active)) break;
        sweep(std::integral_constant<int, 3>{});
        if (!__any(active)) break;
        sweep(std::integral_constant<int, 0>{});
    }
    RG_CTRL_PIN2(ux, uy);
    RG_CTRL_TICK(1);  // sweeps
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
    RG_CTRL_PIN2(v, w);
    RG_CTRL_COUNT_SWEEPS(sweeps);
    RG_CTRL_TICK(2);  // si -> uni, clips
    return my_sweeps;
}

// ------------------------------------------------------------------ neighbour observations
// K nearest neighbours' own-observation rows (staged in LDS) into obs slots 1..K: ascending
// squared distance, ties -> lower index (the canonical order for misc.py:20-25); K >= N-1: all
// others in index order.  Partner rows come from LDS by address, so -- unlike the DPP rounds of the
// controller -- the partners need not be the XOR partners: agent a visits (a + r) mod N, r = 1..N-1.
// With the agent count a compile-time constant (NT) that is N-1 slots and (N-1)(N-2)/2 comparisons
// instead of GW-1 and (GW-1)(GW-2)/2 (N = 5 in groups of 8: 4 and 6 instead of 7 and 21).
template <int GW, int OD, int NT>
__device__ __forceinline__ void write_neighbour_obs(Lds<GW> &lds, int N, int Knb, int ag, int gbase, bool lane_ok,
                                                    float x, float y, float *obs_row) {
    constexpr int M = NT > 0 ? NT - 1 : GW - 1;  // partner slots
    // 64-bit sort keys: (bits of the squared distance, partner index) -- non-negative floats order
    // like their bit patterns, so one unsigned 64-bit compare is the (distance, index) lexicographic
    // test.  Absent partners get keys above every real one.
    unsigned long long key[M > 0 ? M : 1];
    bool ok[M > 0 ? M : 1];
    int rank[M > 0 ? M : 1], who[M > 0 ? M : 1];
    static_for<0, M>([&](auto RR) {
        constexpr int r = decltype(RR)::value;
        int j = ag + r + 1;
        j = j >= N ? j - N : j;
        j = j & (GW - 1);  // idle lanes (ag >= N) stay inside the group's rows
        who[r] = j;
        const float2 pxy = *reinterpret_cast<const float2 *>(&lds.own[gbase + j][0]);  // partner's (x, y)
        const float dx = pxy.x - x, dy = pxy.y - y;
        const float d2 = dx * dx + dy * dy;
        ok[r] = lane_ok & (r + 1 < N);
        const unsigned int hi = ok[r] ? __builtin_bit_cast(unsigned int, d2) : 0xFFFFFFFFu;
        key[r] = (static_cast<unsigned long long>(hi) << 32) | static_cast<unsigned int>(j);
        rank[r] = M - 1 - r;  // pairs in which this slot is the first element; each lost comparison adds one below
    });
    const bool all_others = Knb >= N - 1;
    // rank of a partner = number of partners ahead of it: one comparison per unordered pair (q < k)
    static_for<1, M>([&](auto KK) {
        constexpr int k = decltype(KK)::value;
        static_for<0, k>([&](auto QQ) {
            constexpr int q = decltype(QQ)::value;
            const int q_first = key[q] < key[k] ? 1 : 0;
            rank[k] += q_first;
            rank[q] -= q_first;
        });
    });
    // the rows first (independent LDS reads in flight together), then the predicated stores
    float row[M > 0 ? M : 1][OD];
    static_for<0, M>([&](auto RR) {
        constexpr int r = decltype(RR)::value;
        const float *src = &lds.own[gbase + who[r]][0];
        if constexpr (OD == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src);
            row[r][0] = v.x;
            row[r][1] = v.y;
            row[r][2] = v.z;
            row[r][3] = v.w;
        } else {
#pragma unroll
            for (int c = 0; c < OD; ++c) row[r][c] = src[c];
        }
    });
    static_for<0, M>([&](auto RR) {
        constexpr int r = decltype(RR)::value;
        const int j = who[r];
        const int slot = all_others ? (j < ag ? j : j - 1) : rank[r];
        if (ok[r] & (all_others | (slot < Knb))) {
            float *o = obs_row + (slot + 1) * OD;
            if constexpr (OD == 4) {
                out_store4(o, row[r][0], row[r][1], row[r][2], row[r][3]);
            } else {
#pragma unroll
                for (int c = 0; c < OD; ++c) o[c] = row[r][c];
            }
        }
    });
}

// ------------------------------------------------------------------ the step kernel
// NT: the agent count when it is a compile-time constant (instantiated for GW = 8: 5..8), 0 = read
// it from the parameter block.  One env step of the wave's envs; `sv` = this step's slice of the
// action and output arrays.
// AHEAD: use (and keep filled) the drawn-ahead initial states of rg_state.next_init.  Off in the multi-step launch: there
// the launch costs the MEAN wavefront, and drawing ahead moves the sampler's work without removing any.
// GYM: the gymma block of rg_step_io (gym's TimeLimit + reductions) is compiled in.  Its own instantiations (generic agent
// count, single-step launch): the benchmark kernels carry none of it.
template <int SCN, int GW, bool OBS_ONLY, int NT, bool AHEAD, bool GYM, int QPM = 0, typename QpLds = void>
__device__ __forceinline__ void step_once(const KernelArgs &a, Lds<GW> &lds, const StepView &sv, QpLds *qp_lds = nullptr) {
    constexpr int EPW = WAVE / GW;  // envs per wave
    RG_STAMPS_BEGIN()
    const rg_scenario_params &p = a.p;
    const Consts &k = a.k;
    // ---- kernel arguments first: every pointer and scalar the loads below need, fetched TOGETHER.  Left to itself
    // the compiler fetches each kernarg field where it is first used: five dependent scalar-memory round trips, the
    // state loads issued in between and waited for one group at a time -- 3.3 k of a wave's 24 k cycles went by
    // before the first controller (tools/stamp_probe.py).  The empty asm makes all of them live here: one trip.
    const float *q_poses = a.st.poses, *q_carry = a.st.carry_dist, *q_ret = a.st.ep_return, *q_sum = a.st.done_return_sum;
    const int32_t *q_steps = a.st.episode_steps, *q_act = sv.actions, *q_cnt = a.st.done_count, *q_stp = a.st.done_steps_sum;
    int32_t *q_el = GYM ? sv.io.elapsed : nullptr;  // gymma block (gym TimeLimit's counter)
    const int q_tl = GYM ? sv.io.time_limit : 0;
    const int32_t *q_rc = a.st.reset_count, *q_nep = AHEAD ? a.st.next_episode : nullptr;
    const float *q_nin = AHEAD ? a.st.next_init : nullptr;
    const int q_nst = AHEAD ? a.next_stride : 0;
    const int q_E = a.E, q_epw = a.envs_per_wave, q_P = p.num_prey, q_N = p.n_agents, q_G = gridDim.x;
    asm volatile("" ::"s"(q_poses), "s"(q_carry), "s"(q_ret), "s"(q_sum), "s"(q_steps), "s"(q_act), "s"(q_cnt), "s"(q_stp),
                 "s"(q_E), "s"(q_epw), "s"(q_P), "s"(q_N), "s"(q_G), "s"(q_rc));
    if constexpr (GYM) asm volatile("" ::"s"(q_el), "s"(q_tl));
    if constexpr (AHEAD) asm volatile("" ::"s"(q_nep), "s"(q_nin), "s"(q_nst));
    const float *q_prey = a.st.prey_loc;
    const uint8_t *q_sen = a.st.prey_sensed, *q_cap = a.st.prey_captured, *q_loaded = a.st.loaded, *q_grid = a.st.grid;
    const uint8_t *q_pix = a.st.pixel_type, *q_reached = a.st.reached_goal;
    const int32_t *q_gcol = a.st.goal_col, *q_msg = a.st.messages, *q_zone = a.st.zone_load, *q_load = a.st.load;
    if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) asm volatile("" ::"s"(q_prey), "s"(q_sen), "s"(q_cap));
    else if constexpr (SCN == RG_SCN_WAREHOUSE) asm volatile("" ::"s"(q_loaded));
    else if constexpr (SCN == RG_SCN_SIMPLE) asm volatile("" ::"s"(q_prey));
    else if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) asm volatile("" ::"s"(q_grid), "s"(q_pix), "s"(q_reached), "s"(q_gcol));
    else asm volatile("" ::"s"(q_msg), "s"(q_zone), "s"(q_load));

    const int N = NT > 0 ? NT : q_N;
    const int lane = threadIdx.x;
    const int ag = lane & (GW - 1);
    const int g = lane / GW;
    const int gbase = lane & ~(GW - 1);
    const int epw = q_epw > 0 ? q_epw : EPW;
    const int chunk = xcd_chunk(q_G);
    const int e = chunk * epw + g;
    const bool env_ok = (g < epw) & (e < q_E);
    const bool lane_ok = env_ok && ag < N;
    const size_t eN = static_cast<size_t>(e) * N;

    // ---- loads, ALL issued before anything waits for one of them (coalesced: a wave covers EPW consecutive
    // envs = one contiguous span per array).  What the first controller needs comes first (loads return in
    // order); what only the epilogue needs is fetched raw and decoded there, so its latency hides behind the
    // sub-step loop (RG_LATE pins the first use of such a value to where it is decoded).
#define RG_LATE(v) asm volatile("" : "+v"(v))
    float x = 0.0f, y = 0.0f, th = 0.0f, carry = 0.0f;
    int act = 4;
    int steps_raw = 0;
    float agent_step = 0.0f, sr = 0.0f, cr = 0.0f;
    float st_ret = 0.0f, st_sum = 0.0f;
    int st_cnt = 0, st_steps = 0;
    const bool stats = (!OBS_ONLY) && q_ret != nullptr;
    int pix = 0, reached = 0;  // ArcticTransport (pix feeds this step's goals)
    if (lane_ok) {
        const float *X = q_poses + eN * 3;
        x = X[ag];
        y = X[N + ag];
        th = X[2 * N + ag];
        if constexpr (!OBS_ONLY) act = q_act[eN + ag];
        if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) pix = q_pix[eN + ag];
        agent_step = p.agent_step[ag];
        if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) {
            sr = p.sensing_radius[ag];
            cr = p.capture_radius[ag];
        }
        if constexpr (!OBS_ONLY) carry = q_carry[eN + ag];
    }
    int rc_raw = -1;  // reset_count, for the fused reset of an env that finishes in this launch
    // The env's NEXT initial state, drawn ahead of time into its block of next_init (device_common.h ResetDst) at the
    // end of an earlier launch: fetched now with everything else, so that an env that finishes in this launch -- as a
    // rule the slowest wavefront's, a near-collision env -- starts its next episode with a handful of stores instead of
    // the sampler (Philox, Fisher-Yates through LDS: ~2 k cycles at the very end of the launch's critical path).
    // Only the block's tag is fetched here (4 bytes per env); the block itself is fetched for the few envs that end
    // (load_next below), early enough to hide its latency behind the epilogue.
    const bool ahead = AHEAD && (!OBS_ONLY) && q_nst > 0 && a.auto_reset;
    int nx_tag = -2;                               // episode the block was drawn for
    float nx_pose[3] = {0.0f, 0.0f, 0.0f};         // this lane's agent
    float nx_a = 0.0f, nx_b = 0.0f;                // PCP / Simple: prey `ag` (x, y); MaterialTransport: zone load `ag` (bits)
    uint32_t nx_grid[7] = {0, 0, 0, 0, 0, 0, 0};   // ArcticTransport: this lane's 6 terrain dwords + the goal column
    auto load_next = [&](bool want) {              // this lane's share of the env's drawn-ahead block
        if (want) {
            const float *nb = q_nin + static_cast<size_t>(e) * q_nst;
            if (ag < N) {
                nx_pose[0] = nb[ag];
                nx_pose[1] = nb[N + ag];
                nx_pose[2] = nb[2 * N + ag];
            }
            if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY || SCN == RG_SCN_SIMPLE) {
                if (ag < q_P) {
                    nx_a = nb[3 * N + 2 * ag];
                    nx_b = nb[3 * N + 2 * ag + 1];
                }
            } else if constexpr (SCN == RG_SCN_MATERIAL_TRANSPORT) {
                if (ag < 2) nx_a = nb[3 * N + ag];
            } else if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) {
                const uint32_t *gb = reinterpret_cast<const uint32_t *>(nb + 3 * N);
#pragma unroll
                for (int t = 0; t < 6; ++t) nx_grid[t] = gb[ag * 6 + t];
                nx_grid[6] = gb[24];
            }
        }
    };
    int el_raw = 0;
    if (env_ok) {
        steps_raw = q_steps[e];
        if (!OBS_ONLY && a.auto_reset) rc_raw = q_rc[e];
        if constexpr (GYM) el_raw = q_el[e];
        if (ahead) nx_tag = q_nep[e];
        if (stats && ag == 0) {
            st_ret = q_ret[e];
            st_sum = q_sum[e];
            st_cnt = q_cnt[e];
            st_steps = q_stp[e];
        }
    }
    // scenario state
    uint32_t sen_lo = 0, sen_hi = 0, cap_lo = 0, cap_hi = 0;  // PCP prey flags as bit masks (P <= 64)
    int flag_raw[4] = {0, 0, 0, 0};                          // PCP, P <= 8: this lane's flag bytes, decoded in the epilogue
    int loaded = 0;                                          // Warehouse
    float goal_x = 0.0f, goal_y = 0.0f;                      // Simple
    uint32_t grid_pre[6] = {0, 0, 0, 0, 0, 0};               // ArcticTransport
    int goal_col = 1;
    int load = 0, zone0 = 0, zone1 = 0;                      // MaterialTransport
    int msg[4] = {0, 0, 0, 0};
    // PCP: up to 8 prey (the reference's configurations: 6) go straight into registers now -- every lane
    // of the group loads the same 8 points, one request per group -- so the load latency hides behind
    // the sub-step loop and the epilogue needs no LDS staging; larger prey counts are staged in LDS.
    float2 prey_r[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) prey_r[t] = make_float2(0.0f, 0.0f);
    static_assert(SCN != RG_SCN_ARCTIC_TRANSPORT || GW == 4, "ArcticTransport is a 4-agent scenario");
    if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) {
        const int P = q_P;
        if (P <= 8) {
            if (env_ok) {
                const float2 *src = reinterpret_cast<const float2 *>(q_prey) + static_cast<size_t>(e) * P;
#pragma unroll
                for (int t = 0; t < 8; ++t) prey_r[t] = src[t < P ? t : P - 1];
                if (ag < P) {
                    flag_raw[0] = q_sen[static_cast<size_t>(e) * P + ag];
                    flag_raw[1] = q_cap[static_cast<size_t>(e) * P + ag];
                }
                if constexpr (GW < 8)
                    if (ag + GW < P) {
                        flag_raw[2] = q_sen[static_cast<size_t>(e) * P + ag + GW];
                        flag_raw[3] = q_cap[static_cast<size_t>(e) * P + ag + GW];
                    }
            }
        } else {
            if (env_ok) {
                for (int i = ag; i < 2 * P; i += GW) lds.prey[g][i] = q_prey[static_cast<size_t>(e) * 2 * P + i];
                for (int i = ag; i < P; i += GW) {
                    const uint32_t sb = q_sen[static_cast<size_t>(e) * P + i] != 0;
                    const uint32_t cb = q_cap[static_cast<size_t>(e) * P + i] != 0;
                    if (i < 32) {
                        sen_lo |= sb << i;
                        cap_lo |= cb << i;
                    } else {
                        sen_hi |= sb << (i - 32);
                        cap_hi |= cb << (i - 32);
                    }
                }
            }
            sen_lo = group_or<GW>(sen_lo);
            cap_lo = group_or<GW>(cap_lo);
            if (P > 32) {
                sen_hi = group_or<GW>(sen_hi);
                cap_hi = group_or<GW>(cap_hi);
            }
        }
    } else if constexpr (SCN == RG_SCN_WAREHOUSE) {
        if (lane_ok) loaded = q_loaded[eN + ag];
    } else if constexpr (SCN == RG_SCN_SIMPLE) {
        if (env_ok) {
            goal_x = q_prey[static_cast<size_t>(e) * 2];
            goal_y = q_prey[static_cast<size_t>(e) * 2 + 1];
        }
    } else if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) {
        if (env_ok) {
            // 96 terrain bytes = 24 dwords: 6 per lane of the group (GW = 4)
            const uint32_t *gsrc = reinterpret_cast<const uint32_t *>(q_grid + static_cast<size_t>(e) * 96);
#pragma unroll
            for (int t = 0; t < 6; ++t) grid_pre[t] = gsrc[ag * 6 + t];
            goal_col = q_gcol[e];
        }
        if (lane_ok) reached = q_reached[eN + ag];
    } else {
        if (env_ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // raw: the `% 4` of MaterialTransport.py:119-120 happens in the epilogue
                if (OBS_ONLY || i >= N) msg[i] = q_msg[4 * e + i];
                else msg[i] = q_act[eN + i];
            }
            zone0 = q_zone[2 * e];
            zone1 = q_zone[2 * e + 1];
        }
        if (lane_ok) load = q_load[eN + ag];
    }

    int viol = 0, max_sweeps = 0;
    bool replayed = false;  // wave-uniform: some chunk of this step went through the exact-test replay
    float dist = 0.0f;
    if constexpr (!OBS_ONLY) {
        RG_PIN5(x, y, th, carry, act);
        RG_STAMP(0);  // inputs loaded
        // ---- a1 goal generation (agent.py:48-76, warehouse.py:19-45, MaterialTransport.py:19-46)
        float gx = x, gy = y;
        {
            const int mv = (SCN == RG_SCN_MATERIAL_TRANSPORT) ? act / 4 : act;
            float sd = agent_step;
            if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) {  // agent.py:89-113: drones 0,1; ice 2; water 3
                const float nrm_s = p.arctic_normal_step, slow_s = p.arctic_slow_step, fast_s = p.arctic_fast_step;
                const float water = pix == 1 ? slow_s : pix == 2 ? fast_s : nrm_s;
                const float ice = pix == 1 ? fast_s : pix == 2 ? slow_s : nrm_s;
                sd = ag < 2 ? fast_s : ag == 3 ? water : ice;
            }
            const float cgx = clamp_spec(gx, p.left, p.right), cgy = clamp_spec(gy, p.up, p.down);
            const float lft = (gx - sd) > p.left ? (gx - sd) : p.left;
            const float rgt = (gx + sd) < p.right ? (gx + sd) : p.right;
            const float upw = (gy - sd) > p.up ? (gy - sd) : p.up;
            const float dwn = (gy + sd) < p.down ? (gy + sd) : p.down;
            gx = mv == 0 ? lft : mv == 1 ? rgt : cgx;
            gy = mv == 2 ? upw : mv == 3 ? dwn : cgy;
        }
        // ---- a2 roboEnv.step (utilities/roboEnv.py:52-94), float spec of oracle/oracle_core.h, one
        // CONTROLLER PERIOD (<= 15 sub-steps with v, w held) at a time: theta and dist_travelled
        // advance once per period by fma; inside the period only x, y and (cos, sin) move.
        //
        // _validate every sub-step: the exact test (GW-1 DPP rounds of float math) runs only in a
        // rare wave-uniform branch.  The common path is a conservative pre-test (kernel_args.h): the
        // collision points rounded to binary16 pairs -- per pair round one DPP + v_pk_add_f16 +
        // v_dot2_f32_f16 -- against a threshold widened by the rounding bound, and one boundary test
        // per chunk on the chunk's first position widened by the chunk's travel.  Absent lanes /
        // finished envs sit on far-apart ghost points.  CHUNK sub-steps are advanced and tested
        // together: their test chains are independent, which is the only ILP a single wavefront has here.
        const int thr_pre = __builtin_bit_cast(int, k.thr_pre);  // non-negative floats order like their bit patterns
        const int ghost_q = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pkrtz(1000.0f + 32.0f * ag, -1000.0f));
        float v = 0.0f, w = 0.0f, s = 0.0f, c = 1.0f;
        RG_CHUNK_LOCALS()
        RG_CTRL_LOCALS()
        float acc = carry, last = 0.0f;  // dist incl. the pending sub-step; length of the last sub-step
        bool dead = false;               // group-uniform: the env hit a violation (roboEnv.py:92-94)
        float fin_x = 0.0f, fin_y = 0.0f;
        // x, y = period base (bx, by) + displacement since the period began (ox, oy): a sub-step adds
        // dt*v*(cos, sin) to the small displacement and the position is the single rounding bx + ox, so the
        // roundings of the 29..74 Euler updates do not pile up in x, y (float spec, oracle/oracle_core.h)
        float bx = x, by = y, ox = 0.0f, oy = 0.0f;
        const bool penalize = p.penalize_violations != 0;
        const int U = p.update_frequency, period = p.controller_period;
        for (int it0 = 0; it0 < U; it0 += period) {
            const int n = (U - it0) < period ? (U - it0) : period;
            RG_HEADING_SINCOS(th, s, c);
            const int sw = controller<GW, QPM>(p, k, N, ag, lane_ok, env_ok & !dead, x, y, c, s, gx, gy, v, w, qp_lds RG_CTRL_TICKS_ARG);
            max_sweeps = sw > max_sweeps ? sw : max_sweeps;
            const float dtv = k.dt * v, dtw = k.dt * w;
            float sd, cd;
            sincos_small_spec(dtw, sd, cd);
            if (__any(__builtin_fabsf(dtw) > 0.25f)) {  // only with non-rps time steps / velocity limits
                float sd2, cd2;
                sincos_spec(dtw, sd2, cd2);
                const bool big = __builtin_fabsf(dtw) > 0.25f;
                sd = big ? sd2 : sd;
                cd = big ? cd2 : cd;
            }
            if (it0 == 0) RG_STAMP(1);  // first controller done
            const float mrg = __builtin_fmaf((CHUNK - 1) * 1.000001f, __builtin_fabsf(dtv), PRE_SLACK);
            int n_exec = n;             // sub-steps this env executes in this period
            bool died_now = false;
            // SPARSE collision pre-test (kernel_args.h): v and w are constant inside the period, so between two sub-steps
            // a robot's collision point moves at most |dt v| (the body) + coll_off |dt w| (the chord of its turn) per
            // sub-step.  A test of sub-step u against a threshold widened by the travel of BOTH robots over `span` further
            // sub-steps (2 span M, M = the group's largest per-sub-step travel) covers sub-steps u .. u + span: a pair that
            // passes it cannot be within the collision distance at any of them.  A chunk tests sub-steps 0 (covering 0..2)
            // and 3 (covering the rest) instead of all five; only a chunk that fails goes on to the per-sub-step pre-test
            // and, from there, to the exact replay -- the masks come from the exact test alone, results are unchanged.
            const float mstep = lane_ok ? __builtin_fmaf(__builtin_fabsf(k.coll_off), __builtin_fabsf(dtw), __builtin_fabsf(dtv)) * 1.00001f : 0.0f;
            const float M2 = 2.0f * group_max_nonneg<GW>(mstep);
            auto thr_span = [&](int span) {  // bits of the squared threshold (non-negative floats order like their bit patterns)
                const float t = (k.lin_pre + PRE_SLACK) + static_cast<float>(span) * M2;
                return __builtin_bit_cast(int, (t * t) * 1.00002f);
            };
            const int thr_s1 = thr_span(1), thr_s2 = thr_span(2);
            // the binary16 rounding bound behind lin_pre holds for differences below 0.25 m per axis
            const bool sparse_ok = (k.lin_pre + 2.0f * M2) < 0.24f;

            // C sub-steps starting at sub-step j0 of this period
            auto run_chunk = [&](auto CC, int j0) {
                constexpr int C = decltype(CC)::value;
                const float x0 = x, y0 = y, ox0 = ox, oy0 = oy, c0 = c, s0 = s;
                const bool live = lane_ok & !dead;
                int q[C];
                // every pre-update position of the chunk lies within (C-1)|dt v| of the first
                const bool bnd_any = live & ((__builtin_fabsf(x - k.xc) + mrg > k.xh) | (__builtin_fabsf(y - k.yc) + mrg > k.yh));
                static_for<0, C>([&](auto UU) {
                    constexpr int u = decltype(UU)::value;
                    const float fx = __builtin_fmaf(k.coll_off, c, x), fy = __builtin_fmaf(k.coll_off, s, y);
                    const int qr = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pkrtz(fx, fy));
                    q[u] = live ? qr : ghost_q;
                    // Euler step (Appendix A.4); rotate (cos, sin) by dt*w
                    ox = __builtin_fmaf(c, dtv, ox);
                    oy = __builtin_fmaf(s, dtv, oy);
                    x = bx + ox;
                    y = by + oy;
                    const float cn = __builtin_fmaf(c, cd, -(s * sd));
                    const float sn = __builtin_fmaf(s, cd, c * sd);
                    c = cn;
                    s = sn;
                });
                // smallest squared distance (bits) between this lane's rounded collision point and its partners'
                auto pair_min = [&](int qv) {
                    // the pre-test may visit the pairs in any order.  5 <= N <= 7 in groups of 8: the three quad rounds
                    // cover the pairs inside each quad, then each agent of the upper quad (4 .. N-1) is broadcast over
                    // its quad and mirrored onto the lower one: 3 + (N-4) rounds instead of 7
                    constexpr bool QUAD_COVER = GW == 8 && NT >= 5 && NT <= 7;
                    constexpr int R = QUAD_COVER ? NT - 1 : GW - 1;
                    int d[R];
                    auto diff = [&](int partner_q) {
                        return __builtin_bit_cast(int, __builtin_bit_cast(half2v, qv) - __builtin_bit_cast(half2v, partner_q));
                    };
                    if constexpr (QUAD_COVER) {
                        static_for<1, 4>([&](auto KK) { d[decltype(KK)::value - 1] = diff(xor_lane_i<decltype(KK)::value>(qv)); });
                        static_for<0, NT - 4>([&](auto MM) { d[3 + decltype(MM)::value] = diff(cross_lane_i<decltype(MM)::value>(qv)); });
                    } else {
                        static_for<1, GW>([&](auto KK) { d[decltype(KK)::value - 1] = diff(xor_lane_i<decltype(KK)::value>(qv)); });
                    }
                    int dm = 0x7FFFFFFF;
                    dot2_batch<R>(d);
#pragma unroll
                    for (int r = 0; r < R; ++r) dm = d[r] < dm ? d[r] : dm;
                    return dm;
                };
                // sparse pre-test: sub-step 0 covers 0 .. min(2, C-1), sub-step 3 the rest of the chunk; a chunk that fails
                // adds the remaining sub-steps (the dense pre-test of rounds 1-2: every sub-step against the unwidened
                // threshold), so a failing chunk costs what every chunk used to cost
                int dmin_u[C];  // per sub-step: the replay runs the exact pair test only where the pre-test fired
                bool sparse_hit = live & !sparse_ok;
                {
                    constexpr int SPAN0 = C - 1 < 2 ? C - 1 : 2;
                    const int t0 = SPAN0 == 0 ? thr_pre : SPAN0 == 1 ? thr_s1 : thr_s2;
                    dmin_u[0] = pair_min(q[0]);
                    sparse_hit = sparse_hit | (dmin_u[0] <= t0);
                    if constexpr (C >= 4) {
                        const int t3 = C == 4 ? thr_pre : thr_s1;
                        dmin_u[3] = pair_min(q[3]);
                        sparse_hit = sparse_hit | (dmin_u[3] <= t3);
                    }
                }
                if (penalize && __any(sparse_hit | bnd_any)) {
                RG_CHUNK_DENSE_BEGIN()
                static_for<1, C>([&](auto UU) {
                    constexpr int u = decltype(UU)::value;
                    if constexpr (u != 3) dmin_u[u] = pair_min(q[u]);
                });
                int dmin = dmin_u[0];
                static_for<1, C>([&](auto UU) { dmin = dmin_u[decltype(UU)::value] < dmin ? dmin_u[decltype(UU)::value] : dmin; });
                RG_CHUNK_DENSE_END(dmin)
                if (__any((dmin <= thr_pre) | bnd_any)) {
                    replayed = true;
                    // rare: replay the chunk with the exact float tests of _validate (roboEnv.py:82-94): the
                    // boundary test on every sub-step (per lane, cheap), the pair rounds only on the sub-steps
                    // whose own pre-test fired somewhere in the wave (robots cross the 3 mm pre-test band
                    // within a sub-step or two: most sub-steps of a flagged chunk need no pair rounds)
                    float rx = x0, ry = y0, rox = ox0, roy = oy0, rc = c0, rs = s0;
                    static_for<0, C>([&](auto UU) {
                        constexpr int u = decltype(UU)::value;
                        const bool bnd = lane_ok & !dead & ((rx < k.xmin) | (rx > k.xmax) | (ry < k.ymin) | (ry > k.ymax));
                        const float fx = __builtin_fmaf(k.coll_off, rc, rx), fy = __builtin_fmaf(k.coll_off, rs, ry);
                        bool col = false;
                        if (!__any(dmin_u[u] <= thr_pre)) {
                            // no pair of this sub-step is within the pre-test band: no collision possible
                        } else if constexpr (GW == 8 && NT >= 5 && NT <= 7) {  // same pair cover as the pre-test: 3 + (N-4) rounds
                            static_for<1, 4>([&](auto KK) {
                                constexpr int K = decltype(KK)::value;
                                const float dx = fx - xor_lane<K>(fx), dy = fy - xor_lane<K>(fy);
                                col = col | (((ag ^ K) < N) & (dx * dx + dy * dy <= k.coll_lim2));
                            });
                            static_for<0, NT - 4>([&](auto MM) {  // lower lanes meet agent 4+M, upper lanes agent M
                                constexpr int M = decltype(MM)::value;
                                const float dx = fx - cross_lane<M>(fx), dy = fy - cross_lane<M>(fy);
                                col = col | (dx * dx + dy * dy <= k.coll_lim2);
                            });
                        } else {
                            static_for<1, GW>([&](auto KK) {
                                constexpr int K = decltype(KK)::value;
                                const float dx = fx - xor_lane<K>(fx), dy = fy - xor_lane<K>(fy);
                                col = col | (((ag ^ K) < N) & (dx * dx + dy * dy <= k.coll_lim2));
                            });
                        }
                        col = col & lane_ok & !dead;
                        const int code = (group_any<GW>(col, gbase) ? 1 : 0) | (group_any<GW>(bnd, gbase) ? 2 : 0);
                        rox = __builtin_fmaf(rc, dtv, rox);  // the violating sub-step is still integrated
                        roy = __builtin_fmaf(rs, dtv, roy);
                        rx = bx + rox;
                        ry = by + roy;
                        if (env_ok & !dead & (code != 0)) {
                            viol = code;
                            n_exec = j0 + u + 1;
                            died_now = true;
                            dead = true;
                            fin_x = rx;
                            fin_y = ry;
                        }
                        const float cn = __builtin_fmaf(rc, cd, -(rs * sd));
                        const float sn = __builtin_fmaf(rs, cd, rc * sd);
                        rc = cn;
                        rs = sn;
                    });
                    RG_CHUNK_REPLAY_END(rx, ry)
                }
                }
            };
            int j = 0;
            for (; j + CHUNK <= n; j += CHUNK) run_chunk(std::integral_constant<int, CHUNK>{}, j);
            static_for<1, CHUNK>([&](auto RR) {  // the remainder as one shorter chunk
                if (n - j == decltype(RR)::value) run_chunk(RR, j);
            });

            // period end: heading and distance for the sub-steps this env executed
            const bool upd = env_ok & (!dead | died_now);
            const float ne = static_cast<float>(n_exec);
            const float adv = __builtin_fabsf(dtv);
            th = upd ? wrap_spec(__builtin_fmaf(ne, dtw, th)) : th;
            acc = upd ? __builtin_fmaf(ne, adv, acc) : acc;
            last = upd ? adv : last;
            {   // base <- base + displacement (= x, y as last formed); displacement <- the exact remainder (TwoSum)
                const float tx = x - bx, ty = y - by;
                ox = (bx - (x - tx)) + (ox - tx);
                oy = (by - (y - ty)) + (oy - ty);
                bx = x;
                by = y;
            }
            if (it0 == 0) RG_STAMP(2);  // first period done
            if (!__any(env_ok & !dead)) break;
        }
        if (dead) {
            x = fin_x;
            y = fin_y;
        }
        dist = viol ? acc : acc - last;
        carry = last;
        RG_STAMP(3);  // all periods done
        RG_CTRL_REPORT()
        RG_CHUNK_REPORT()
    }

    // ---- scenario epilogue
    const int D = p.obs_dim;
    float *obs_row = sv.io.obs + (eN + ag) * D;
    bool done = false;
    int remaining = -1;
    float reward = 0.0f;
    RG_LATE(steps_raw);
    const int steps = steps_raw + (OBS_ONLY ? 0 : 1);
    // an env that ends by a violation or by the step limit (all but a few endings) is known to end already: its
    // drawn-ahead block is fetched now, behind the epilogue
    RG_LATE(rc_raw);
    RG_LATE(nx_tag);
    const bool have_next = ahead & (nx_tag == rc_raw);  // the block holds exactly the episode that would start now
    const bool next_early = env_ok & have_next & ((viol != 0) | (steps > p.max_episode_steps));
    if constexpr (AHEAD) load_next(next_early);

    if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) {
        const int P = q_P;
        const float sr2 = sr * sr, cr2 = cr * cr;
        if (P > 8) __syncthreads();  // LDS prey block visible (single-wave workgroup: waitcnt + s_barrier)
        if (P <= 8) {  // the flag bytes fetched in the prologue -> bit masks of the env
            RG_LATE(flag_raw[0]);
            RG_LATE(flag_raw[1]);
            uint32_t sb = (flag_raw[0] != 0 ? 1u : 0u) << ag, cb = (flag_raw[1] != 0 ? 1u : 0u) << ag;
            if constexpr (GW < 8) {
                RG_LATE(flag_raw[2]);
                RG_LATE(flag_raw[3]);
                sb |= (flag_raw[2] != 0 ? 1u : 0u) << (ag + GW);
                cb |= (flag_raw[3] != 0 ? 1u : 0u) << (ag + GW);
            }
            sen_lo = group_or<GW>(sb);
            cap_lo = group_or<GW>(cb);
        }
        uint32_t nsen_lo = sen_lo, nsen_hi = sen_hi, ncap_lo = cap_lo, ncap_hi = cap_hi;
        // The prey block is scanned four at a time (LDS reads in flight together).  scan(lo, hi, f)
        // calls f(i, prey_x, prey_y, d2) for i in [lo, hi).
        auto scan = [&](int lo, int hi, auto &&f) {
            for (int i0 = lo; i0 < hi; i0 += 4) {
                float2 pl[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int i = (i0 + t) < hi ? (i0 + t) : (hi - 1);
                    pl[t] = *reinterpret_cast<const float2 *>(&lds.prey[g][2 * i]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float dx = x - pl[t].x, dy = y - pl[t].y;
                    f(i0 + t, (i0 + t) < hi, pl[t].x, pl[t].y, dx * dx + dy * dy);
                }
            }
        };
        float closest = -1.0f, qx = -5.0f, qy = -5.0f;
        if (P <= 8) {
            // common case (P = 6): the whole prey block in registers, one pass for tracking and
            // the nearest-prey search, no second trip to LDS
            const float2 (&pl)[8] = prey_r;
            float d2[8];
            uint32_t s_b = 0, c_b = 0;
            const bool acts = lane_ok & (act == 4);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float dx = x - pl[t].x, dy = y - pl[t].y;
                d2[t] = dx * dx + dy * dy;
                const bool in = t < P;
                s_b |= ((in & lane_ok & (d2[t] <= sr2)) ? 1u : 0u) << t;
                c_b |= ((in & acts & (d2[t] <= cr2)) ? 1u : 0u) << t;
            }
            if constexpr (!OBS_ONLY) {  // a11 _update_tracking_and_locations (PredatorCapturePrey.py:72-95)
                s_b = group_or<GW>(s_b);
                c_b = group_or<GW>(c_b);
                nsen_lo = sen_lo | (s_b & ~cap_lo);            // sensed: any agent in range, prey not yet captured
                ncap_lo = cap_lo | (nsen_lo & c_b & ~cap_lo);  // captured: sensed and a 'no_action' agent in range
                if (env_ok && ag < P) {
                    a.st.prey_sensed[static_cast<size_t>(e) * P + ag] = (nsen_lo >> ag) & 1u;
                    a.st.prey_captured[static_cast<size_t>(e) * P + ag] = (ncap_lo >> ag) & 1u;
                }
                if constexpr (GW < 8)
                    if (env_ok && ag + GW < P) {
                        a.st.prey_sensed[static_cast<size_t>(e) * P + ag + GW] = (nsen_lo >> (ag + GW)) & 1u;
                        a.st.prey_captured[static_cast<size_t>(e) * P + ag + GW] = (ncap_lo >> (ag + GW)) & 1u;
                    }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {  // a13 nearest uncaptured prey within the agent's own sensing radius
                const bool cap = ((ncap_lo >> t) & 1u) != 0;
                const bool take = (t < P) & !cap & (d2[t] <= sr2) & ((d2[t] < closest) | (closest == -1.0f));
                qx = take ? pl[t].x : qx;
                qy = take ? pl[t].y : qy;
                closest = take ? d2[t] : closest;
            }
        } else {
        if constexpr (!OBS_ONLY) {  // a11, general P
            uint32_t s_lo = 0, s_hi = 0, c_lo = 0, c_hi = 0;  // prey this agent senses / could capture
            const bool acts = lane_ok & (act == 4);
            const int P32 = P < 32 ? P : 32;
            scan(0, P32, [&](int i, bool in, float, float, float d2) {
                s_lo |= ((in & lane_ok & (d2 <= sr2)) ? 1u : 0u) << (i & 31);
                c_lo |= ((in & acts & (d2 <= cr2)) ? 1u : 0u) << (i & 31);
            });
            s_lo = group_or<GW>(s_lo);
            c_lo = group_or<GW>(c_lo);
            nsen_lo = sen_lo | (s_lo & ~cap_lo);
            ncap_lo = cap_lo | (nsen_lo & c_lo & ~cap_lo);
            if (P > 32) {
                scan(32, P, [&](int i, bool in, float, float, float d2) {
                    s_hi |= ((in & lane_ok & (d2 <= sr2)) ? 1u : 0u) << ((i - 32) & 31);
                    c_hi |= ((in & acts & (d2 <= cr2)) ? 1u : 0u) << ((i - 32) & 31);
                });
                s_hi = group_or<GW>(s_hi);
                c_hi = group_or<GW>(c_hi);
                nsen_hi = sen_hi | (s_hi & ~cap_hi);
                ncap_hi = cap_hi | (nsen_hi & c_hi & ~cap_hi);
            }
            if (env_ok) {
                for (int i = ag; i < P; i += GW) {
                    const uint32_t sw_ = i < 32 ? nsen_lo >> i : nsen_hi >> (i - 32);
                    const uint32_t cw_ = i < 32 ? ncap_lo >> i : ncap_hi >> (i - 32);
                    a.st.prey_sensed[static_cast<size_t>(e) * P + i] = sw_ & 1u;
                    a.st.prey_captured[static_cast<size_t>(e) * P + i] = cw_ & 1u;
                }
            }
        }
        {   // a13, general P
            auto nearest = [&](uint32_t capmask, int base) {
                return [&, capmask, base](int i, bool in, float px, float py, float d2) {
                    const bool cap = ((capmask >> ((i - base) & 31)) & 1u) != 0;
                    const bool take = in & !cap & (d2 <= sr2) & ((d2 < closest) | (closest == -1.0f));
                    qx = take ? px : qx;
                    qy = take ? py : qy;
                    closest = take ? d2 : closest;
                };
            };
            scan(0, P < 32 ? P : 32, nearest(ncap_lo, 0));
            if (P > 32) scan(32, P, nearest(ncap_hi, 32));
        }
        }
        RG_STAMP_E(0);  // prey tracked, nearest prey found
        const int od = p.capability_aware ? 6 : 4;
        lds.own[lane][0] = x;
        lds.own[lane][1] = y;
        lds.own[lane][2] = qx;
        lds.own[lane][3] = qy;
        lds.own[lane][4] = sr;
        lds.own[lane][5] = cr;
        __syncthreads();
        if (od == 6) {
            if (lane_ok) {
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) obs_row[cc] = lds.own[lane][cc];
            }
            write_neighbour_obs<GW, 6, NT>(lds, N, p.num_neighbors, ag, gbase, lane_ok, x, y, obs_row);
        } else {
            if (lane_ok) out_store4(obs_row, x, y, qx, qy);
            write_neighbour_obs<GW, 4, NT>(lds, N, p.num_neighbors, ag, gbase, lane_ok, x, y, obs_row);
        }
        RG_STAMP_E(1);  // observations written
        if constexpr (!OBS_ONLY) {  // a14 reward / termination (PredatorCapturePrey.py:155-176, 209-216)
            const int unseen0 = P - __builtin_popcount(sen_lo) - __builtin_popcount(sen_hi);
            const int left0 = P - __builtin_popcount(cap_lo) - __builtin_popcount(cap_hi);
            const int unseen1 = P - __builtin_popcount(nsen_lo) - __builtin_popcount(nsen_hi);
            const int left1 = P - __builtin_popcount(ncap_lo) - __builtin_popcount(ncap_hi);
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
    } else if constexpr (SCN == RG_SCN_WAREHOUSE) {  // a15 (warehouse.py:102-178): obs BEFORE the reward mutates `loaded`
        lds.own[lane][0] = x;
        lds.own[lane][1] = y;
        lds.own[lane][2] = loaded ? 1.0f : 0.0f;
        __syncthreads();
        if (lane_ok) {
            obs_row[0] = x;
            obs_row[1] = y;
            obs_row[2] = loaded ? 1.0f : 0.0f;
        }
        write_neighbour_obs<GW, 3, NT>(lds, N, p.num_neighbors, ag, gbase, lane_ok, x, y, obs_row);
        if constexpr (!OBS_ONLY) {
            if (viol) {
                reward = p.violation_reward;
                done = true;
            } else {
                const bool green = (ag % 2) == 0;  // warehouse.py:63-65
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
                if (lane_ok) a.st.loaded[eN + ag] = static_cast<uint8_t>(loaded);
            }
        }
    } else if constexpr (SCN == RG_SCN_SIMPLE) {  // scenarios/Simple/simple.py:155-225
        lds.own[lane][0] = x;
        lds.own[lane][1] = y;
        __syncthreads();
        if (lane_ok) {
            obs_row[0] = x;
            obs_row[1] = y;
            obs_row[2 * N] = goal_x;
            obs_row[2 * N + 1] = goal_y;
        }
        write_neighbour_obs<GW, 2, NT>(lds, N, N - 1, ag, gbase, lane_ok, x, y, obs_row);  // all others, index order
        if constexpr (!OBS_ONLY) {
            if (viol) {
                reward = p.violation_reward;
                done = true;
            } else {
                const float dx = x - goal_x, dy = y - goal_y;
                const float r = -(dx * dx + dy * dy);
                reward = r * p.reward_scaler;
                done = steps > p.max_episode_steps;
            }
        }
    } else if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) {  // ArcticTransport.py:84-143, agent.py:14-87
        {
            uint32_t *gdst = reinterpret_cast<uint32_t *>(&lds.grid[g][0]);
#pragma unroll
            for (int t = 0; t < 6; ++t) gdst[ag * 6 + t] = grid_pre[t];
        }
        __syncthreads();
        const uint8_t *grid = &lds.grid[g][0];
        // get_cell_from_pose: int() truncates toward zero; /0.25 is exact
        int row = -static_cast<int>((y - 1.0f) / 0.25f), col = static_cast<int>((x + 1.5f) / 0.25f);
        row = row < 0 ? 0 : row > 7 ? 7 : row;
        col = col < 0 ? 0 : col > 11 ? 11 : col;
        if constexpr (!OBS_ONLY) {
            pix = grid[row * 12 + col];
            reached = reached | (pix == 3 ? 1 : 0);
        }
        const int here = grid[row * 12 + col];
        lds.own[lane][0] = x;
        lds.own[lane][1] = y;
        lds.own[lane][2] = static_cast<float>(here);
        lds.aload[lane] = row * 16 + col;
        __syncthreads();
        const float goalx = static_cast<float>(goal_col) * 0.25f - 1.5f, goaly = -1.0f * 0.25f + 0.75f;
        if (lane_ok) {
            obs_row[0] = x;
            obs_row[1] = y;
            obs_row[2] = static_cast<float>(here);
            // the other three in the order of agent.py:42-69: {1,2,3} {0,2,3} {3,0,1} {2,0,1}
            const int o0 = ag == 0 ? 1 : ag == 1 ? 0 : ag == 2 ? 3 : 2;
            const int o1 = ag < 2 ? 2 : 0, o2 = ag < 2 ? 3 : 1;
            const int oth[3] = {o0, o1, o2};
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                obs_row[3 + 3 * m + 0] = lds.own[gbase + oth[m]][0];
                obs_row[3 + 3 * m + 1] = lds.own[gbase + oth[m]][1];
                obs_row[3 + 3 * m + 2] = lds.own[gbase + oth[m]][2];
            }
            obs_row[12] = goalx;
            obs_row[13] = goaly;
#pragma unroll
            for (int i = 0; i < 2; ++i) {  // the 8 cells around each drone, edges clamped
                const int rc = lds.aload[gbase + i];
                const int r_ = rc >> 4, c_ = rc & 15;
                const int left = c_ > 0 ? c_ - 1 : c_, right = c_ < 11 ? c_ + 1 : c_;
                const int up = r_ > 0 ? r_ - 1 : r_, down = r_ < 7 ? r_ + 1 : r_;
                float *o = obs_row + 14 + 8 * i;
                o[0] = static_cast<float>(grid[up * 12 + left]);
                o[1] = static_cast<float>(grid[r_ * 12 + left]);
                o[2] = static_cast<float>(grid[down * 12 + left]);
                o[3] = static_cast<float>(grid[up * 12 + c_]);
                o[4] = static_cast<float>(grid[down * 12 + c_]);
                o[5] = static_cast<float>(grid[up * 12 + right]);
                o[6] = static_cast<float>(grid[r_ * 12 + right]);
                o[7] = static_cast<float>(grid[down * 12 + right]);
            }
        }
        if constexpr (!OBS_ONLY) {
            // shared reward over the two ground robots, in agent order (ArcticTransport.py:125-134)
            const float dx = x - goalx, dy = y - goaly;
            lds.ax[lane] = dx * dx + dy * dy;
            lds.ay[lane] = static_cast<float>(pix * 2 + reached);
            __syncthreads();
            if (viol) {
                reward = p.violation_reward;
                done = true;
            } else {
                reward = 0.0f;
                bool all_reached = true;
#pragma unroll
                for (int j = 2; j < 4; ++j) {
                    const int pr = static_cast<int>(lds.ay[gbase + j]);
                    const bool rj = (pr & 1) != 0;
                    if (!rj) reward = reward + p.not_reached_penalty;
                    if ((pr >> 1) != 3) reward = reward + p.dist_multiplier * lds.ax[gbase + j];
                    all_reached = all_reached && rj;
                }
                done = steps > p.max_episode_steps;
                if (!done) done = all_reached;
            }
            if (lane_ok) {
                a.st.pixel_type[eN + ag] = static_cast<uint8_t>(pix);
                a.st.reached_goal[eN + ag] = static_cast<uint8_t>(reached);
            }
        }
    } else {  // a16 MaterialTransport (MaterialTransport.py:113-189)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            RG_LATE(msg[i]);
            if (!OBS_ONLY && i < N) msg[i] = msg[i] % 4;  // MaterialTransport.py:119-120
        }
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
                obs_row[10] = agent_step;
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
                const float zr2 = p.zone1_radius * p.zone1_radius;
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
                        } else if (jx * jx + jy * jy <= zr2) {
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

    RG_STAMP(4);  // scenario epilogue computed
    if constexpr (!OBS_ONLY) {
        // gym's TimeLimit on top of the scenario (gymma block of rg_step_io): the limit ends an episode the scenario did not
        constexpr bool gym = GYM;
        bool trunc = false;
        if (gym) {
            RG_LATE(el_raw);
            trunc = env_ok & !done & (el_raw + 1 >= q_tl);
        }
        const bool ended = done | trunc;
        // sum of the agents' rewards in agent order (read when shared_reward == 0, and by the gymma block)
        float rsum = 0.0f;
        if ((stats && !p.shared_reward) || gym) {
            __syncthreads();
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
            out_store1(sv.io.reward + eN + ag, reward);
            out_store1(sv.io.dist_travelled + eN + ag, dist);
            if (ag == 0) {
                a.st.episode_steps[e] = steps;
                if (stats) {  // misc.py:178-185: episodeReward += reward[0] | sum(reward)
                    RG_LATE(st_ret);
                    RG_LATE(st_sum);
                    RG_LATE(st_cnt);
                    RG_LATE(st_steps);
                    float ret = st_ret + (p.shared_reward ? reward : rsum);
                    if (ended) {  // a truncated episode counts like a finished one (run_env counts them, misc.py:186-206)
                        a.st.done_return_sum[e] = st_sum + ret;
                        a.st.done_count[e] = st_cnt + 1;
                        a.st.done_steps_sum[e] = st_steps + steps;
                        ret = 0.0f;
                    }
                    a.st.ep_return[e] = ret;
                }
                sv.io.done[e] = done ? 1 : 0;
                if (gym) {
                    q_el[e] = ended ? 0 : el_raw + 1;
                    sv.io.truncated[e] = trunc ? 1 : 0;
                    sv.io.ended[e] = ended ? 1 : 0;
                    sv.io.reward_sum[e] = rsum;
                }
                sv.io.violation[e] = static_cast<uint8_t>(viol);
                sv.io.remaining[e] = remaining;
                if (sv.io.qp_sweeps) sv.io.qp_sweeps[e] = max_sweeps;
            }
            if constexpr (GYM) {
                // rg_step_io.zero_obs_on_end: an env that ends hands the trainer the reset observation (zeros).  This lane wrote
                // its agent's row above (its own block and its neighbours' slots); the same lane's later stores win.
                if (sv.io.zero_obs_on_end && ended) {
                    for (int c = 0; c < D; ++c) obs_row[c] = 0.0f;
                }
            }
        }
        RG_STAMP(5);  // outputs stored
        // ---- fused auto-reset of finished envs (scenario.reset(); ~1 env in 70 per step)
        // an env whose block holds exactly the episode that starts now copies it; any other runs the sampler
        if (a.auto_reset && __any(env_ok & ended)) {
            __syncthreads();  // the wave's state stores are issued before the resetting lanes rewrite them
            if constexpr (AHEAD) load_next(env_ok & ended & have_next & !next_early);  // ended some other way: fetched late
            if (__any(env_ok & ended & !have_next)) reset_group<SCN, GW>(a, lds, e, g, ag, env_ok & ended & !have_next, rc_raw);
            if (env_ok & ended & have_next) {  // the same stores reset_group makes with commit = true
                if (ag < N) {
                    float *X = a.st.poses + eN * 3;
                    X[ag] = nx_pose[0];
                    X[N + ag] = nx_pose[1];
                    X[2 * N + ag] = nx_pose[2];
                    a.st.carry_dist[eN + ag] = 0.0f;
                    if constexpr (SCN == RG_SCN_WAREHOUSE) a.st.loaded[eN + ag] = 0;
                    if constexpr (SCN == RG_SCN_MATERIAL_TRANSPORT) {
                        a.st.load[eN + ag] = 0;
                        if (ag < 4) a.st.messages[4 * e + ag] = 0;
                    }
                    if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) {
                        a.st.pixel_type[eN + ag] = 0;
                        a.st.reached_goal[eN + ag] = 0;
                    }
                }
                if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY || SCN == RG_SCN_SIMPLE) {
                    const int P = q_P;
                    const float *nb = q_nin + static_cast<size_t>(e) * q_nst + 3 * N;
                    for (int i = ag; i < P; i += GW) {  // prey `ag` was prefetched; more than GW prey: the rest from the block
                        float *pl = a.st.prey_loc + (static_cast<size_t>(e) * P + i) * 2;
                        pl[0] = i == ag ? nx_a : nb[2 * i];
                        pl[1] = i == ag ? nx_b : nb[2 * i + 1];
                        if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) {
                            a.st.prey_sensed[static_cast<size_t>(e) * P + i] = 0;
                            a.st.prey_captured[static_cast<size_t>(e) * P + i] = 0;
                        }
                    }
                } else if constexpr (SCN == RG_SCN_MATERIAL_TRANSPORT) {
                    if (ag < 2) a.st.zone_load[2 * e + ag] = __builtin_bit_cast(int32_t, nx_a);
                } else if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) {
                    uint32_t *gd = reinterpret_cast<uint32_t *>(a.st.grid + static_cast<size_t>(e) * 96);
#pragma unroll
                    for (int t = 0; t < 6; ++t) gd[ag * 6 + t] = nx_grid[t];
                    if (ag == 0) a.st.goal_col[e] = static_cast<int32_t>(nx_grid[6]);
                }
                if (ag == 0) {
                    a.st.reset_count[e] = rc_raw + 1;
                    a.st.episode_steps[e] = 0;
                }
            }
        }
        // ---- draw ahead: an env that goes on and whose block does not hold its next episode (consumed by the reset of
        // an earlier launch, or never drawn) gets it now.  The block of an env that finished in THIS launch is left
        // stale on purpose (it is redrawn in a later launch, off this launch's critical path).
        // A wavefront that is already long -- a QP of one of its envs needed more than two sweeps, or a chunk was
        // replayed -- leaves the draw to a later launch (an env that finishes before it happened runs the sampler as
        // before): the draw must not lengthen the waves the launch is waiting for.
        if (AHEAD && ahead && !(replayed | __any(max_sweeps > 2)) && __any(env_ok & !ended & !have_next)) {
            const bool need = env_ok & !ended & !have_next;
            __syncthreads();  // (LDS scratch of a reset above is free again)
            reset_group<SCN, GW>(a, lds, e, g, ag, need, rc_raw, reset_dst_next(a, e));
            if (need && ag == 0) a.st.next_episode[e] = rc_raw;
        }
        RG_STAMP(6);  // reset done
        RG_STAMPS_WRITE(lane, sv.io.qp_sweeps, chunk * EPW, a.E, max_sweeps)
    }
}

// ROLLOUT = false: one env step per launch (rg_step, rg_get_obs).  ROLLOUT = true (rg_rollout): the
// wave's envs advance num_steps times with no device-wide synchronisation between steps; state
// round-trips through this CU's caches (workgroup-scope visibility after the barrier).  Separate
// instantiations: the loop keeps more values live (at N = 5 the thread-per-env kernel goes from 237 to
// 313 VGPRs) and would slow the single-step launch down.
// QPM = RG_QP_CVXOPT: the certificate's QP by the interior-point iteration of ipm_qp.h.  Its own instantiations (generic agent
// count): the launch is then dominated by that iteration (about ten times the rest of the step), and it needs one wave per SIMD.
template <int SCN, int GW, bool OBS_ONLY, int NT, bool ROLLOUT, bool GYM = false, int QPM = 0>
__global__ __launch_bounds__(WAVE) void step_kernel(const KernelArgs a) {
    __shared__ Lds<GW> lds;
    const int N = NT > 0 ? NT : a.p.n_agents;
    if constexpr (QPM == RG_QP_CVXOPT) {
        static_assert(GW == 4 || GW == 8, "the interior-point mode admits n_agents <= 8");
        __shared__ ipm::GroupLds<GW> qp_lds;   // records + one workspace per env (ipm_qp.h): these instantiations only
        if constexpr (!ROLLOUT) {
            step_once<SCN, GW, OBS_ONLY, NT, true, GYM, QPM>(a, lds, step_view(a, 0, N, a.p.obs_dim), &qp_lds);
        } else {
            for (int t = 0; t < a.num_steps; ++t) {
                if (t) __syncthreads();
                step_once<SCN, GW, OBS_ONLY, NT, false, false, QPM>(a, lds, step_view(a, t, N, a.p.obs_dim), &qp_lds);
            }
        }
    } else if constexpr (!ROLLOUT) {
        step_once<SCN, GW, OBS_ONLY, NT, true, GYM, QPM>(a, lds, step_view(a, 0, N, a.p.obs_dim), static_cast<void *>(nullptr));
    } else {
        for (int t = 0; t < a.num_steps; ++t) {
            if (t) __syncthreads();
            step_once<SCN, GW, OBS_ONLY, NT, false, false, QPM>(a, lds, step_view(a, t, N, a.p.obs_dim), static_cast<void *>(nullptr));
        }
    }
}

template <int SCN, int GW>
__global__ __launch_bounds__(WAVE) void reset_kernel(const KernelArgs a) {
    constexpr int EPW = WAVE / GW;
    __shared__ Lds<GW> lds;
    const int lane = threadIdx.x;
    const int ag = lane & (GW - 1);
    const int g = lane / GW;
    const int e = blockIdx.x * EPW + g;
    const bool env_ok = e < a.E;
    const bool want = env_ok && (a.reset_mask == nullptr || a.reset_mask[e] != 0);
    if (!__any(want)) return;
    // the running return restarts with the episode; RG_RESET_BOOK_EPISODE first books the abandoned
    // episode as finished (an episode cut short by a time limit outside the scenario)
    const bool stats = want && ag == 0 && a.st.ep_return != nullptr;
    const int steps_before = stats ? a.st.episode_steps[e] : 0;
    reset_group<SCN, GW>(a, lds, e, g, ag, want);
    if (stats) {
        if ((a.reset_flags & RG_RESET_BOOK_EPISODE) && steps_before > 0) {
            a.st.done_return_sum[e] = a.st.done_return_sum[e] + a.st.ep_return[e];
            a.st.done_count[e] = a.st.done_count[e] + 1;
            a.st.done_steps_sum[e] = a.st.done_steps_sum[e] + steps_before;
        }
        a.st.ep_return[e] = 0.0f;
    }
}

}  // namespace rg


// ------------------------------------------------------------------ host side: launch dispatch
namespace rg {

inline int group_width(int N) { return N <= 4 ? 4 : N <= 8 ? 8 : 16; }

template <int SCN, bool OBS_ONLY, bool ROLLOUT>
static hipError_t launch_step_scn(const KernelArgs &a_in, hipStream_t stream) {
    const int gw = group_width(a_in.p.n_agents);
    // A batch that leaves SIMDs idle with full wavefronts runs with partly filled ones instead (more
    // waves, each carrying fewer envs, as long as there is at most one wave per SIMD: 1024): a wave's
    // time is the maximum over its envs (QP sweeps, replays, resets), and the idle SIMDs are free.
    KernelArgs a = a_in;
    int epw = WAVE / gw;
    if constexpr (!kStampsBuild) {   // (a stamps build keeps full waves: its eight slots are the wave's first eight envs)
        while (epw >= 2 && (a.E + epw / 2 - 1) / (epw / 2) <= RG_MAX_WAVES) epw /= 2;
        a.envs_per_wave = epw;
    }
    const int grid = (a.E + epw - 1) / epw;
    if constexpr (!OBS_ONLY) {
        // the interior-point mode's kernels live in their own translation units (robogym_kernels_ipm.hip, robogym_rollout_group_ipm.hip)
        if (a.p.qp_mode == RG_QP_CVXOPT) return ROLLOUT ? launch_rollout_ipm(a, grid, stream) : launch_step_ipm(a, grid, stream);
    }
    if constexpr (!OBS_ONLY && !ROLLOUT) {
        if (a.io.elapsed) {  // gymma block: its own instantiations (generic agent count)
            if (gw == 4) hipLaunchKernelGGL((step_kernel<SCN, 4, false, 0, false, true>), dim3(grid), dim3(WAVE), 0, stream, a);
            else if (gw == 16) hipLaunchKernelGGL((step_kernel<SCN, 16, false, 0, false, true>), dim3(grid), dim3(WAVE), 0, stream, a);
            else hipLaunchKernelGGL((step_kernel<SCN, 8, false, 0, false, true>), dim3(grid), dim3(WAVE), 0, stream, a);
            return hipGetLastError();
        }
    }
    if (gw == 4) hipLaunchKernelGGL((step_kernel<SCN, 4, OBS_ONLY, 0, ROLLOUT>), dim3(grid), dim3(WAVE), 0, stream, a);
    else if (gw == 16) hipLaunchKernelGGL((step_kernel<SCN, 16, OBS_ONLY, 0, ROLLOUT>), dim3(grid), dim3(WAVE), 0, stream, a);
    else if constexpr (OBS_ONLY) hipLaunchKernelGGL((step_kernel<SCN, 8, true, 0, false>), dim3(grid), dim3(WAVE), 0, stream, a);
    else if (a.p.n_agents == 5) hipLaunchKernelGGL((step_kernel<SCN, 8, false, 5, ROLLOUT>), dim3(grid), dim3(WAVE), 0, stream, a);
    else if (a.p.n_agents == 6) hipLaunchKernelGGL((step_kernel<SCN, 8, false, 6, ROLLOUT>), dim3(grid), dim3(WAVE), 0, stream, a);
    else if (a.p.n_agents == 7) hipLaunchKernelGGL((step_kernel<SCN, 8, false, 7, ROLLOUT>), dim3(grid), dim3(WAVE), 0, stream, a);
    else hipLaunchKernelGGL((step_kernel<SCN, 8, false, 8, ROLLOUT>), dim3(grid), dim3(WAVE), 0, stream, a);
    return hipGetLastError();
}

// the lane-group step for every scenario; OBS_ONLY exists for ROLLOUT = false only
template <bool OBS_ONLY, bool ROLLOUT>
static hipError_t launch_step_group(const KernelArgs &a, hipStream_t stream) {
    static_assert(!(OBS_ONLY && ROLLOUT), "rg_get_obs is a single launch");
    switch (a.p.scenario) {
        case RG_SCN_PREDATOR_CAPTURE_PREY:
            return launch_step_scn<RG_SCN_PREDATOR_CAPTURE_PREY, OBS_ONLY, ROLLOUT>(a, stream);
        case RG_SCN_WAREHOUSE:
            return launch_step_scn<RG_SCN_WAREHOUSE, OBS_ONLY, ROLLOUT>(a, stream);
        case RG_SCN_MATERIAL_TRANSPORT:
            return launch_step_scn<RG_SCN_MATERIAL_TRANSPORT, OBS_ONLY, ROLLOUT>(a, stream);
        case RG_SCN_SIMPLE:
            return launch_step_scn<RG_SCN_SIMPLE, OBS_ONLY, ROLLOUT>(a, stream);
        case RG_SCN_ARCTIC_TRANSPORT:
            if constexpr (!OBS_ONLY) {
                if (a.p.qp_mode == RG_QP_CVXOPT)
                    return ROLLOUT ? launch_rollout_ipm(a, (a.E + 15) / 16, stream) : launch_step_ipm(a, (a.E + 15) / 16, stream);
            }
            if constexpr (!OBS_ONLY && !ROLLOUT) {
                if (a.io.elapsed) {
                    hipLaunchKernelGGL((step_kernel<RG_SCN_ARCTIC_TRANSPORT, 4, false, 0, false, true>), dim3((a.E + 15) / 16),
                                       dim3(WAVE), 0, stream, a);
                    return hipGetLastError();
                }
            }
            hipLaunchKernelGGL((step_kernel<RG_SCN_ARCTIC_TRANSPORT, 4, OBS_ONLY, 0, ROLLOUT>), dim3((a.E + 15) / 16), dim3(WAVE), 0,
                               stream, a);
            return hipGetLastError();
        default:
            return hipErrorInvalidValue;
    }
}

// QPM = RG_QP_CVXOPT: launch dispatch of the interior-point mode's kernels (rg_create admits n_agents <= 8 in this mode).  Called
// from the translation units that instantiate them, compiled with their own scheduler flag (build.py FILE_FLAGS).
template <int SCN, bool ROLLOUT>
static hipError_t launch_ipm_scn(const KernelArgs &a, int grid, hipStream_t stream) {
    constexpr int Q = RG_QP_CVXOPT;
    const int gw = SCN == RG_SCN_ARCTIC_TRANSPORT ? 4 : group_width(a.p.n_agents);
    if constexpr (!ROLLOUT) {
        if (a.io.elapsed) {   // gymma block
            if (gw == 4) hipLaunchKernelGGL((step_kernel<SCN, 4, false, 0, false, true, Q>), dim3(grid), dim3(WAVE), 0, stream, a);
            else if constexpr (SCN != RG_SCN_ARCTIC_TRANSPORT) hipLaunchKernelGGL((step_kernel<SCN, 8, false, 0, false, true, Q>), dim3(grid), dim3(WAVE), 0, stream, a);
            return hipGetLastError();
        }
    }
    if (gw == 4) hipLaunchKernelGGL((step_kernel<SCN, 4, false, 0, ROLLOUT, false, Q>), dim3(grid), dim3(WAVE), 0, stream, a);
    else if constexpr (SCN != RG_SCN_ARCTIC_TRANSPORT) hipLaunchKernelGGL((step_kernel<SCN, 8, false, 0, ROLLOUT, false, Q>), dim3(grid), dim3(WAVE), 0, stream, a);
    return hipGetLastError();
}
template <bool ROLLOUT>
static hipError_t launch_ipm_group(const KernelArgs &a, int grid, hipStream_t stream) {
    switch (a.p.scenario) {
        case RG_SCN_PREDATOR_CAPTURE_PREY: return launch_ipm_scn<RG_SCN_PREDATOR_CAPTURE_PREY, ROLLOUT>(a, grid, stream);
        case RG_SCN_WAREHOUSE: return launch_ipm_scn<RG_SCN_WAREHOUSE, ROLLOUT>(a, grid, stream);
        case RG_SCN_MATERIAL_TRANSPORT: return launch_ipm_scn<RG_SCN_MATERIAL_TRANSPORT, ROLLOUT>(a, grid, stream);
        case RG_SCN_SIMPLE: return launch_ipm_scn<RG_SCN_SIMPLE, ROLLOUT>(a, grid, stream);
        case RG_SCN_ARCTIC_TRANSPORT: return launch_ipm_scn<RG_SCN_ARCTIC_TRANSPORT, ROLLOUT>(a, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace rg
