// step_tpe.h -- the env-step kernel, THREAD-PER-ENV mapping (gfx950 / MI355X, CDNA4).
// (Templates only; instantiated by robogym_tpe.hip for rg_step and robogym_rollout_tpe.hip for rg_rollout.)
//
// Same arithmetic, same outputs, bit for bit, as the lane-group kernel of robogym_kernels.hip
// (both reproduce the float spec of oracle/oracle_core.h); only the mapping differs:
//
//   lane-group kernel : GW lanes per env, one lane per agent, pair work by DPP rounds.  Shortest
//                       chain for a handful of envs (E < ~1k), but 3 of 8 lanes idle at N = 5, every
//                       pair is computed on both of its lanes, and group-uniform work runs on every
//                       lane: ~8 800 lane-instructions per agent-step.
//   thread-per-env    : one lane owns a whole env; all N agents and N(N-1)/2 pairs live in that
//     (this file)       lane's registers (N is a template parameter, every loop is unrolled, every
//                       array index is a compile-time constant).  No cross-lane traffic at all, each
//                       pair computed once, N independent agent chains per lane for ILP:
//                       ~1 600 lane-instructions per agent-step.  One wave = 64 envs.
//
// With one wave per SIMD up to 65 536 envs the launch time is flat (one wave's chain), beyond that
// the kernel is VALU-issue bound.  Registers are not a constraint at <= 1 wave per SIMD (512 VGPRs).
// The host picks this kernel for N <= 6 and E >= tpe_min_envs (robogym_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "device_common.h"
#include "probes/diag.h"   // diagnostic hooks: every RG_* macro below expands to nothing in the shipped build

namespace rg {
namespace tpe {

constexpr int CH = 5;  // sub-steps per validated chunk (controller periods of 15 and 14 sub-steps: 5+5+5, 5+5+4)

template <int I, int END, typename F>
__device__ __forceinline__ void sfor(F &&f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, END>(f);
    }
}

template <int N>
constexpr int group_width() {
    return N <= 2 ? 2 : N <= 4 ? 4 : N <= 8 ? 8 : 16;
}

// visits the unordered pairs (i, j), i < j < N, in the XOR 1-factorisation order of the spec
// (k = 1..GW-1, i ascending, j = i ^ k): f(integral_constant i, integral_constant j)
template <int N, typename F>
__device__ __forceinline__ void for_pairs(F &&f) {
    constexpr int GW = group_width<N>();
    sfor<1, GW>([&](auto KK) {
        constexpr int K = decltype(KK)::value;
        sfor<0, N>([&](auto II) {
            constexpr int I = decltype(II)::value;
            constexpr int J = I ^ K;
            if constexpr (I < J && J < N) f(II, std::integral_constant<int, J>{});
        });
    });
}

// sums of the per-agent values pa, pb over the agents [LO, LO + W) present (< N) in the order of the lane-group kernel's butterfly
// (oracle_core.h barrier_qp): block(LO, W) = block(LO, W/2) + block(LO + W/2, W/2); a right half of absent agents adds nothing
template <int LO, int W, int N>
__device__ __forceinline__ void block_sums(const float (&pa)[N], const float (&pb)[N], float &sa, float &sb) {
    if constexpr (W == 1) {
        sa = pa[LO];
        sb = pb[LO];
    } else {
        block_sums<LO, W / 2, N>(pa, pb, sa, sb);
        if constexpr (LO + W / 2 < N) {
            float ra, rb;
            block_sums<LO + W / 2, W / 2, N>(pa, pb, ra, rb);
            sa = sa + ra;
            sb = sb + rb;
        }
    }
}
// RG_QP_CVXOPT in this kernel: the lane's env, its N records (xi, thresholded uhat) in the lane's slice of the wave's LDS block,
// through the interior-point iteration of ipm_qp.h (rows in registers: one lane computes every row).  Defined in step_tpe_ipm.h,
// which the instantiation units include -- the host simulation of tests/sanitize/ never instantiates this mode.
template <int N>
__device__ int ipm_lane(const rg_scenario_params &p, const float (&xix)[N], const float (&xiy)[N], float (&ux)[N], float (&uy)[N],
                        float *rec);

// ------------------------------------------------------------------ controller (a3..a8)
// QPM: include/robogym.h RG_QP_*; rec: the lane's LDS slice for the records of the interior-point mode
template <int N, int QPM = 0>
__device__ __forceinline__ int controller(const rg_scenario_params &p, const Consts &k, const float (&x)[N],
                                          const float (&y)[N], const float (&c)[N], const float (&s)[N],
                                          const float (&gx)[N], const float (&gy)[N], float (&v)[N], float (&w)[N], float *rec = nullptr) {
    float xix[N], xiy[N], ux[N], uy[N], uhx[N], uhy[N];
#pragma unroll
    for (int a = 0; a < N; ++a) {  // a4 uni_to_si_states, a5 si_position_controller
        xix[a] = x[a] + k.pd * c[a];
        xiy[a] = y[a] + k.pd * s[a];
        float dx = gx[a] - xix[a], dy = gy[a] - xiy[a];
        const float nrm = norm2_spec(dx, dy);
        const float sc = k.pvl / nrm;
        const bool clip = nrm > k.pvl;
        ux[a] = clip ? dx * sc : dx;
        uy[a] = clip ? dy * sc : dy;
    }
    if constexpr (QPM == RG_QP_CVXOPT) {
        // a6 as the reference's stack evaluates it (oracle/oracle_core.h barrier_qp_ipm_spec): threshold, interior-point iterate
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const float n2u = ux[a] * ux[a] + uy[a] * uy[a];
            if (n2u > k.bml * k.bml) {
                const float sc = k.bml / __builtin_sqrtf(n2u);
                ux[a] = ux[a] * sc;
                uy[a] = uy[a] * sc;
            }
        }
        const int iters = ipm_lane<N>(p, xix, xiy, ux, uy, rec);
#pragma unroll
        for (int a = 0; a < N; ++a) {  // a7 si_to_uni_dyn, a8 set_velocities
            float vv = c[a] * ux[a] + s[a] * uy[a];
            float ww = k.inv_pd * (-s[a] * ux[a] + c[a] * uy[a]);
            ww = ww > k.wlim ? k.wlim : ww;
            ww = ww < -k.wlim ? -k.wlim : ww;
            vv = vv > k.vmax ? k.vmax : vv;
            vv = vv < -k.vmax ? -k.vmax : vv;
            ww = ww > k.wmax ? k.wmax : ww;
            ww = ww < -k.wmax ? -k.wmax : ww;
            v[a] = vv;
            w[a] = ww;
        }
        return iters;
    }
    // a6 barrier certificate (oracle/oracle_core.h barrier_qp): pair constants
    const float bgain = p.barrier_gain, ugain = p.unsafe_barrier_gain, qp_rtol = p.qp_rtol;
    const int qp_cap = p.qp_max_sweeps;
    const bool has_unsafe = p.barrier_has_unsafe_gain != 0;
    float ex[N][N], ey[N][N], fx[N][N], fy[N][N], bp[N][N], emax[N][N], mu[N][N], muA[N][N], muB[N][N];
    for_pairs<N>([&](auto II, auto JJ) {
        constexpr int i = decltype(II)::value, j = decltype(JJ)::value;
        const float dx = xix[i] - xix[j], dy = xiy[i] - xiy[j];
        const float ee = dx * dx + dy * dy;
        const float h = ee - k.r2;
        const float gain = ((h >= 0.0f) | !has_unsafe) ? bgain : ugain;
        const float b = gain * ((h * h) * h);
        const float n2 = 2.0f * ee;
        const bool ok = n2 > 0.0f;
        const float rn2 = ok ? 1.0f / n2 : 0.0f;
        ex[i][j] = dx;
        ey[i][j] = dy;
        fx[i][j] = dx * rn2;
        fy[i][j] = dy * rn2;
        bp[i][j] = (0.5f * b) * rn2;
        emax[i][j] = ok ? fmaxf(__builtin_fabsf(dx), __builtin_fabsf(dy)) : 0.0f;
        mu[i][j] = muA[i][j] = muB[i][j] = 0.0f;
    });
#pragma unroll
    for (int a = 0; a < N; ++a) {  // "Threshold control inputs before QP", decided on squares
        const float n2u = ux[a] * ux[a] + uy[a] * uy[a];
        if (n2u > k.bml * k.bml) {
            const float sc = k.bml / __builtin_sqrtf(n2u);
            ux[a] = ux[a] * sc;
            uy[a] = uy[a] * sc;
        }
        uhx[a] = ux[a];
        uhy[a] = uy[a];
    }
    int sweeps = 0;
    bool active = true;
    auto sweep = [&](auto PH) {
        constexpr int PHASE = decltype(PH)::value;
        float chg = 0.0f;
        float pa[N], pb[N];
#pragma unroll
        for (int a_ = 0; a_ < N; ++a_) pa[a_] = pb[a_] = 0.0f;
        for_pairs<N>([&](auto II, auto JJ) {
            constexpr int i = decltype(II)::value, j = decltype(JJ)::value;
            const float c0 = mu[i][j] - bp[i][j];
            const float t = __builtin_fmaf(fy[i][j], uy[j] - uy[i], c0);
            float mn = __builtin_fmaf(fx[i][j], ux[j] - ux[i], t);
            mn = (mn > 0.0f) ? mn : 0.0f;
            const float delta = mn - mu[i][j];
            mu[i][j] = mn;
            ux[i] = __builtin_fmaf(delta, ex[i][j], ux[i]);
            uy[i] = __builtin_fmaf(delta, ey[i][j], uy[i]);
            ux[j] = __builtin_fmaf(-delta, ex[i][j], ux[j]);
            uy[j] = __builtin_fmaf(-delta, ey[i][j], uy[j]);
            chg = fmaxf(chg, __builtin_fabsf(delta) * emax[i][j]);
            if constexpr (PHASE == 1) muA[i][j] = mn;
            if constexpr (PHASE == 2) muB[i][j] = mn;
            // restart bookkeeping: the two inner products of the extrapolation factor, per agent over its partners in round order
            // (for_pairs visits the pairs round by round), formed while the third sweep of a block runs -- as a pass of its own
            // after the sweep the N = 6 instantiation spilled twice as many values (196 scratch instructions against 94, 20 % slower)
            if constexpr (PHASE == 3) {
                const float d1 = muB[i][j] - muA[i][j], d2 = mn - muB[i][j];
                const float dd = d2 - d1;
                pa[i] = __builtin_fmaf(dd, d2, pa[i]);
                pb[i] = __builtin_fmaf(dd, dd, pb[i]);
                pa[j] = __builtin_fmaf(dd, d2, pa[j]);
                pb[j] = __builtin_fmaf(dd, dd, pb[j]);
            }
        });
        ++sweeps;
        float umax = k.bml;
#pragma unroll
        for (int a = 0; a < N; ++a) umax = fmaxf(umax, fmaxf(__builtin_fabsf(ux[a]), __builtin_fabsf(uy[a])));
        active = (chg > qp_rtol * umax) & (sweeps < qp_cap);
        if constexpr (PHASE == 3) {
            if (active) {  // restart (oracle_core.h barrier_qp): one extrapolation factor for all multipliers; u rebuilt agent by agent
                constexpr int GW = group_width<N>();
                // <dd, d2> and <dd, dd> per agent over its partners in round order, then in the order of the lane-group kernel's
                // butterfly ((s0 + s1) + (s2 + s3)) + ...; agents >= N would add exact zeros and are left out
                float ga, gb;
                block_sums<0, GW, N>(pa, pb, ga, gb);
                const bool ok = (gb > 0.0f) & (ga < 0.0f) & (-ga < 32.0f * gb);
                const float gam = ok ? ga / gb : 0.0f;
                for_pairs<N>([&](auto II, auto JJ) {
                    constexpr int i = decltype(II)::value, j = decltype(JJ)::value;
                    float m = __builtin_fmaf(-gam, mu[i][j] - muB[i][j], mu[i][j]);
                    m = (m > 0.0f) ? m : 0.0f;
                    mu[i][j] = m;
                });
                sfor<0, N>([&](auto AA) {
                    constexpr int a = decltype(AA)::value;
                    float sx = uhx[a], sy = uhy[a];
                    sfor<1, GW>([&](auto KK) {
                        constexpr int q = a ^ decltype(KK)::value;
                        if constexpr (q < N) {
                            constexpr int lo = a < q ? a : q, hi = a < q ? q : a;
                            // absent pairs (emax == 0) carry mu == 0: adding 0 * e is exact
                            const float sgx = a < q ? ex[lo][hi] : -ex[lo][hi], sgy = a < q ? ey[lo][hi] : -ey[lo][hi];
                            if (emax[lo][hi] > 0.0f) {
                                sx = __builtin_fmaf(mu[lo][hi], sgx, sx);
                                sy = __builtin_fmaf(mu[lo][hi], sgy, sy);
                            }
                        }
                    });
                    ux[a] = sx;
                    uy[a] = sy;
                });
            }
        }
    };
    while (active) {
        sweep(std::integral_constant<int, 1>{});
        if (!active) break;
        sweep(std::integral_constant<int, 2>{});
        if (!active) break;
        sweep(std::integral_constant<int, 3>{});
        if (!active) break;
        sweep(std::integral_constant<int, 0>{});
    }
#pragma unroll
    for (int a = 0; a < N; ++a) {  // a7 si_to_uni_dyn, a8 set_velocities
        float vv = c[a] * ux[a] + s[a] * uy[a];
        float ww = k.inv_pd * (-s[a] * ux[a] + c[a] * uy[a]);
        ww = ww > k.wlim ? k.wlim : ww;
        ww = ww < -k.wlim ? -k.wlim : ww;
        vv = vv > k.vmax ? k.vmax : vv;
        vv = vv < -k.vmax ? -k.vmax : vv;
        ww = ww > k.wmax ? k.wmax : ww;
        ww = ww < -k.wmax ? -k.wmax : ww;
        v[a] = vv;
        w[a] = ww;
    }
    return sweeps;
}

// ------------------------------------------------------------------ coalesced stores through LDS
// One lane owns one env, so a lane's own stores are strided by the env's record: a wave-level store instruction
// touches 64 cache lines and becomes up to 64 write requests to L2 (measured at a chip-filling batch: 52.7 M write
// requests per launch for 231 MB, TCC busy 91 %).  The wave's 64 records are contiguous in memory, so each lane writes
// its record to the wave's LDS block instead and the wave copies the block out with lane-contiguous 16-byte stores
// (16 lines per instruction).  One wave per workgroup and all 64 lanes take part (lanes past the end of the batch
// repeat its last env, see step_kernel): LDS operations of a wave execute in program order, no barrier is involved;
// the fences only pin the compiler's order.
// 16 KB: eight one-wave workgroups per CU (two waves per SIMD) fit the CU's 160 KB; 12 KB for the instantiations that
// run three waves per SIMD (N <= 4: twelve workgroups per CU)
// (RG_TPE_W4 / W5 / W78: waves per SIMD the instantiations are compiled for; defaults in probes/diag.h)
constexpr int tpe_waves(int n) { return n <= 4 ? RG_TPE_W4 : n == 5 ? RG_TPE_W5 : n == 6 ? 2 : RG_TPE_W78; }
// the wave's LDS block in floats: as many one-wave workgroups as the SIMDs' wave slots must fit a CU's 160 KB
template <int N>
constexpr int stage_dw() { return tpe_waves(N) >= 4 ? 2560 : tpe_waves(N) == 3 ? 3072 : 4096; }
typedef float f4v __attribute__((ext_vector_type(4), may_alias));
typedef float f2v __attribute__((ext_vector_type(2), may_alias));

struct Stage {
    float *buf;   // the wave's LDS block, stage_dw<N>() floats
    int lane;     // lane = env slot of the wave
    int nact;     // envs of this wave (64 except in the batch's last wave)
    size_t env0;  // first env of the wave
    size_t e;     // this lane's env (lanes past the end of the batch: its last env)
    int *flag;    // -DRG_TPE_GUARD builds only (tests/guard_probe.py): a store outside its array is dropped and flagged in
    int E;        // done_count[0] instead of faulting (RG_GUARDED, probes/diag.h); dead in the shipped build
};

__device__ __forceinline__ void stage_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// g[0 .. ndw) <- buf[0 .. ndw): the wave's contiguous span of one array of REC floats per env (ndw = nact * REC).
// All LDS reads of the span are issued before the first store (one LDS round trip, not one per store).
template <int REC>
__device__ __forceinline__ void copy_span(const Stage &sg, const float *buf, float *g, int ndw, const float *lo = nullptr,
                                          const float *hi = nullptr) {
    if (((ndw | static_cast<int>(reinterpret_cast<uintptr_t>(g) >> 2)) & 3) == 0) {
        constexpr int IT = (REC + 3) / 4;  // 64 REC floats = 16 REC units of 16 bytes, 64 units per instruction
        f4v t[IT];
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int u = 4 * sg.lane + 4 * WAVE * q;
            t[q] = *reinterpret_cast<const f4v *>(buf + (u < ndw ? u : 0));
        }
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int u = 4 * sg.lane + 4 * WAVE * q;
            if (u < ndw) {
                RG_GUARDED(g + u, lo, hi - 3, 1 << 8, *reinterpret_cast<f4v *>(g + u) = t[q]);
            }
        }
    } else {  // the batch's last wave, or a base that is not 16-byte aligned
        for (int u = sg.lane; u < ndw; u += WAVE) {
            RG_GUARDED(g + u, lo, hi, 2 << 8, g[u] = buf[u]);
        }
    }
}

// for env in [0, nact): g[env * gs + (0 .. run)) <- buf[env * run + (0 .. run)) (the LDS block holds the runs back to
// back), in units of W dwords (run, gs multiples of W; at most 64 units per run).  One iteration moves 64 / upr whole
// runs: a lane keeps its place (env within the iteration, unit within the run) and only the wave-uniform env base
// advances.  Four iterations at a time: their LDS reads are issued together, then their stores.
template <int W>
__device__ __forceinline__ void copy_runs(const Stage &sg, const float *buf, float *g, int run, int gs, const float *lo = nullptr,
                                          const float *hi = nullptr) {
    typedef float unit_t __attribute__((ext_vector_type(W == 1 ? 1 : W), may_alias));
    constexpr int UNR = 4;
    const int upr = run / W;  // units per run (wave-uniform)
    const float rcp = __builtin_amdgcn_rcpf(static_cast<float>(upr));
    // small non-negative integers: floor((n + 0.5) / upr) is exact through the approximate reciprocal
    const int epi = __builtin_amdgcn_readfirstlane(static_cast<int>(64.5f * rcp));  // envs per iteration
    const int le = static_cast<int>((static_cast<float>(sg.lane) + 0.5f) * rcp);    // this lane's env in the iteration
    const int r = sg.lane - le * upr;                                               // and its unit in the run
    const unsigned dst0 = static_cast<unsigned>(le * gs + r * W);
    const int last = le < epi ? sg.nact - le : 0;  // this lane moves the iterations whose first env is below `last`
    for (int e0 = 0; e0 < sg.nact; e0 += UNR * epi) {
        unit_t t[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int eq = e0 + q * epi;
            t[q] = *reinterpret_cast<const unit_t *>(buf + (eq < last ? (eq * upr + sg.lane) * W : 0));
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int eq = e0 + q * epi;
            if (eq < last) {
                float *dst = g + static_cast<size_t>(eq) * gs + dst0;
                RG_GUARDED(dst, lo, hi - (W - 1), 4 << 8, *reinterpret_cast<unit_t *>(dst) = t[q]);
            }
        }
    }
}

// ------------------------------------------------------------------ neighbour rows
// obs slots 1..K of agent A from the own-observation rows of the K nearest others (ascending
// squared distance, ties -> lower index); K >= N-1: all others in index order.
template <int N, int OD, int A>
__device__ __forceinline__ void write_neighbours(const float (&x)[N], const float (&y)[N], const float (&own)[N][OD],
                                                 int Knb, float *obs_row) {
    unsigned long long key[N];
    int rank[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float dx = x[j] - x[A], dy = y[j] - y[A];
        key[j] = (static_cast<unsigned long long>(__builtin_bit_cast(unsigned int, dx * dx + dy * dy)) << 32) |
                 static_cast<unsigned int>(j);
        rank[j] = 0;
    }
    const bool all_others = Knb >= N - 1;
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int q = 0; q < j; ++q) {
            if (j == A || q == A) continue;
            const int q_first = key[q] < key[j] ? 1 : 0;
            rank[j] += q_first;
            rank[q] += 1 - q_first;
        }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (j == A) continue;
        const int slot = all_others ? (j < A ? j : j - 1) : rank[j];
        if (all_others | (slot < Knb)) {
            float *o = obs_row + (slot + 1) * OD;
            if constexpr (OD == 4) {
                *reinterpret_cast<float4 *>(o) = make_float4(own[j][0], own[j][1], own[j][2], own[j][3]);
            } else {
#pragma unroll
                for (int cc = 0; cc < OD; ++cc) o[cc] = own[j][cc];
            }
        }
    }
}

// The observation rows of the wave's envs: row A of this lane's env is written by fn(integral_constant A, row
// pointer) -- every element of the D floats.  When rows are whole 16-byte units (D a multiple of 4: the benchmark
// formats) they go through the wave's LDS block in batches of RP rows of all 64 envs, RP fixed at compile time from
// DMAX, the largest D the instantiation can see, each batch copied out as 64 runs of rows x D floats.  Otherwise the
// copy would move 4 bytes per lane and cost more LDS round trips than the stores it saves: the lane writes its rows
// to memory itself.
template <int N, int DMAX, typename F>
__device__ __forceinline__ void stage_obs_rows(const Stage &sg, float *obs, int D, F &&fn) {
    constexpr int RP = (stage_dw<N>() / (WAVE * DMAX)) < N ? (stage_dw<N>() / (WAVE * DMAX)) : N;
    static_assert(RP >= 1, "one observation row of the whole wave must fit the staging block");
    // D > DMAX: a parameter block that asks for more neighbour slots than there are other agents (rows wider than the
    // agents fill, misc.py:20-25 / PredatorCapturePrey.py:198-201) -- the batch would overrun the staging block
    if ((D & 3) != 0 || D > DMAX) {
        float *mine = obs + sg.e * N * D;
        sfor<0, N>([&](auto AA) { fn(AA, mine + decltype(AA)::value * D); });
        return;
    }
    float *g0 = obs + sg.env0 * N * D;  // the wave's first env
    sfor<0, N>([&](auto AA) {
        constexpr int A = decltype(AA)::value;
        constexpr int A0 = A - A % RP, CNT = (A0 + RP <= N) ? RP : N - A0;  // this row's batch: rows A0 .. A0 + CNT
        fn(AA, sg.buf + (sg.lane * CNT + (A - A0)) * D);
        if constexpr (A == A0 + CNT - 1) {
            stage_fence();
            const float *lo = kTpeGuard ? obs : nullptr, *hi = kTpeGuard ? obs + static_cast<size_t>(sg.E) * N * D : nullptr;
            copy_runs<4>(sg, sg.buf, g0 + A0 * D, CNT * D, N * D, lo, hi);
            stage_fence();
        }
    });
}

// own row + the rows of the Knb nearest others (PredatorCapturePrey, Warehouse, Simple: D >= OD (Knb + 1); `tail`
// writes whatever follows the neighbour slots)
template <int N, int OD, int DMAX, typename T>
__device__ __forceinline__ void write_obs_staged(const float (&x)[N], const float (&y)[N], const float (&own)[N][OD],
                                                 int Knb, const Stage &sg, float *obs, int D, T &&tail) {
    stage_obs_rows<N, DMAX>(sg, obs, D, [&](auto AA, float *row) {
        constexpr int A = decltype(AA)::value;
        if constexpr (OD == 4) {
            *reinterpret_cast<f4v *>(row) = f4v{own[A][0], own[A][1], own[A][2], own[A][3]};
        } else {
#pragma unroll
            for (int cc = 0; cc < OD; ++cc) row[cc] = own[A][cc];
        }
        write_neighbours<N, OD, A>(x, y, own, Knb, row);
        tail(row);
    });
}

// ------------------------------------------------------------------ one env step on one lane
// returns whether the episode ended (roboEnv.py:38-96 + the scenario's step())
template <int SCN, int N, int QPM = 0>
__device__ __forceinline__ bool step_env(const KernelArgs &a, const StepView &sv, const int e, int &rc_raw, const Stage &sg) {
    RG_STAMPS_BEGIN()
    const rg_scenario_params &p = a.p;
    const Consts &k = a.k;
    const size_t eN = static_cast<size_t>(e) * N;

    // ---- kernel arguments the loads need, fetched together (one scalar-memory round trip instead of one per
    // field at its first use; see step_group.h), then ALL the loads before anything waits for one of them
    const float *q_poses = a.st.poses, *q_carry = a.st.carry_dist, *q_ret = a.st.ep_return, *q_sum = a.st.done_return_sum;
    const int32_t *q_steps = a.st.episode_steps, *q_act = sv.actions, *q_cnt = a.st.done_count, *q_stp = a.st.done_steps_sum;
    if constexpr (!kHostSim)   // (scheduling hint only; the host simulation of tests/sanitize/ has no "s" registers)
        asm volatile("" ::"s"(q_poses), "s"(q_carry), "s"(q_ret), "s"(q_sum), "s"(q_steps), "s"(q_act), "s"(q_cnt), "s"(q_stp));
    float x[N], y[N], th[N], acc[N], last[N];
    int act[N];
    int pix[N];
    {
        const float *X = q_poses + eN * 3;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            x[i] = X[i];
            y[i] = X[N + i];
            th[i] = X[2 * N + i];
            act[i] = q_act[eN + i];
            last[i] = 0.0f;
        }
        if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) {
#pragma unroll
            for (int i = 0; i < N; ++i) pix[i] = a.st.pixel_type[eN + i];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = q_carry[eN + i];  // dist incl. the pending sub-step (first used at the period end)
    }
    RG_PIN_ARR5(N, x, y, th, act, acc);
    RG_TSTAMP_MAIN(0);  // inputs loaded
    // (step counter, reset counter and statistics words are fetched in the epilogue, where they are used: this kernel
    // has no register to spare -- a value more live through the step costs its second wave per SIMD -- and the second
    // wave hides the latency)

    const bool stats = q_ret != nullptr;

    // ---- a1 goal generation
    float gx[N], gy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int mv = (SCN == RG_SCN_MATERIAL_TRANSPORT) ? act[i] / 4 : act[i];
        float sd = p.agent_step[i];
        if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) {
            const float nrm_s = p.arctic_normal_step, slow_s = p.arctic_slow_step, fast_s = p.arctic_fast_step;
            const float water = pix[i] == 1 ? slow_s : pix[i] == 2 ? fast_s : nrm_s;
            const float ice = pix[i] == 1 ? fast_s : pix[i] == 2 ? slow_s : nrm_s;
            sd = i < 2 ? fast_s : i == 3 ? water : ice;
        }
        const float cgx = clamp_spec(x[i], p.left, p.right), cgy = clamp_spec(y[i], p.up, p.down);
        const float lft = (x[i] - sd) > p.left ? (x[i] - sd) : p.left;
        const float rgt = (x[i] + sd) < p.right ? (x[i] + sd) : p.right;
        const float upw = (y[i] - sd) > p.up ? (y[i] - sd) : p.up;
        const float dwn = (y[i] + sd) < p.down ? (y[i] + sd) : p.down;
        gx[i] = mv == 0 ? lft : mv == 1 ? rgt : cgx;
        gy[i] = mv == 2 ? upw : mv == 3 ? dwn : cgy;
    }

    // ---- a2 roboEnv.step, one controller period at a time (float spec of oracle/oracle_core.h)
    int viol = 0, max_sweeps = 0;
    RG_TPE_DIAG_LOCALS()
    const bool penalize = p.penalize_violations != 0;
    const int U = p.update_frequency, period = p.controller_period;
    const int thr_pre = __builtin_bit_cast(int, k.thr_pre);  // non-negative floats order like their bit patterns
    float v[N], w[N], s[N], c[N];
    // x, y = period base (bx, by) + displacement since the period began (ox, oy); see step_group.h
    float bx[N], by[N], ox[N], oy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        bx[i] = x[i];
        by[i] = y[i];
        ox[i] = 0.0f;
        oy[i] = 0.0f;
    }
    for (int it0 = 0; it0 < U && !viol; it0 += period) {
        const int n = (U - it0) < period ? (U - it0) : period;
#pragma unroll
        for (int i = 0; i < N; ++i) sincos_spec(th[i], s[i], c[i]);
        // (interior-point mode: the records go through the lane's slice of the staging block, which is idle until the stores)
        const int sw = controller<N, QPM>(p, k, x, y, c, s, gx, gy, v, w, QPM ? sg.buf + sg.lane * (4 * N) : nullptr);
        max_sweeps = sw > max_sweeps ? sw : max_sweeps;
        RG_PIN_ARR2(N, v, w);
        if (it0 == 0) RG_TSTAMP_MAIN(1);  // first controller
        RG_TPE_DIAG_SWEEPS(sw, it0);
        float dtv[N], dtw[N], sd[N], cd[N], mrg[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            dtv[i] = k.dt * v[i];
            dtw[i] = k.dt * w[i];
            mrg[i] = __builtin_fmaf((CH - 1) * 1.000001f, __builtin_fabsf(dtv[i]), PRE_SLACK);
            if (__builtin_fabsf(dtw[i]) <= 0.25f) sincos_small_spec(dtw[i], sd[i], cd[i]);
            else sincos_spec(dtw[i], sd[i], cd[i]);
        }
        int n_exec = n;
        // exact _validate on the current (pre-update) poses: bit 0 collision, bit 1 boundary
        auto validate = [&](const float (&vx)[N], const float (&vy)[N], const float (&vc)[N], const float (&vs)[N]) {
            int code = 0;
            float fx[N], fy[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if ((vx[i] < k.xmin) | (vx[i] > k.xmax) | (vy[i] < k.ymin) | (vy[i] > k.ymax)) code |= 2;
                fx[i] = __builtin_fmaf(k.coll_off, vc[i], vx[i]);
                fy[i] = __builtin_fmaf(k.coll_off, vs[i], vy[i]);
            }
            for_pairs<N>([&](auto II, auto JJ) {
                constexpr int i = decltype(II)::value, j = decltype(JJ)::value;
                const float dx = fx[i] - fx[j], dy = fy[i] - fy[j];
                if (dx * dx + dy * dy <= k.coll_lim2) code |= 1;
            });
            return code;
        };
        auto advance = [&](float (&ax)[N], float (&ay)[N], float (&aox)[N], float (&aoy)[N], float (&ac)[N], float (&as)[N]) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                aox[i] = __builtin_fmaf(ac[i], dtv[i], aox[i]);
                aoy[i] = __builtin_fmaf(as[i], dtv[i], aoy[i]);
                ax[i] = bx[i] + aox[i];
                ay[i] = by[i] + aoy[i];
                const float cn = __builtin_fmaf(ac[i], cd[i], -(as[i] * sd[i]));
                const float sn = __builtin_fmaf(as[i], cd[i], ac[i] * sd[i]);
                ac[i] = cn;
                as[i] = sn;
            }
        };
        // C sub-steps starting at sub-step j0; returns false when the env hit a violation
        auto run_chunk = [&](auto CC, int j0) -> bool {
            constexpr int C = decltype(CC)::value;
            float ox0[N], oy0[N], c0[N], s0[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                ox0[i] = ox[i];
                oy0[i] = oy[i];
                c0[i] = c[i];
                s0[i] = s[i];
            }
            // conservative pre-tests (kernel_args.h): one boundary test per chunk on its first positions
            // widened by the chunk's travel; per sub-step the collision points rounded to binary16 pairs
            int dmin = 0x7FFFFFFF;
            float bmax_x = 0.0f, bmax_y = 0.0f;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                bmax_x = fmaxf(bmax_x, __builtin_fabsf(x[i] - k.xc) + mrg[i]);
                bmax_y = fmaxf(bmax_y, __builtin_fabsf(y[i] - k.yc) + mrg[i]);
            }
            const bool bnd_any = (bmax_x > k.xh) | (bmax_y > k.yh);
            sfor<0, C>([&](auto UU) {
                half2v q[N];
#pragma unroll
                for (int i = 0; i < N; ++i)
                    q[i] = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf(k.coll_off, c[i], x[i]), __builtin_fmaf(k.coll_off, s[i], y[i])));
                int dd[N * (N - 1) / 2];  // the sub-step's pair differences, squared as one hazard-safe batch (device_common.h)
                int np = 0;
                for_pairs<N>([&](auto II, auto JJ) {
                    constexpr int i = decltype(II)::value, j = decltype(JJ)::value;
                    dd[np++] = __builtin_bit_cast(int, q[i] - q[j]);
                });
                dot2_batch<N * (N - 1) / 2>(dd);
#pragma unroll
                for (int r = 0; r < N * (N - 1) / 2; ++r) dmin = dd[r] < dmin ? dd[r] : dmin;
                advance(x, y, ox, oy, c, s);
            });
            if (penalize && ((dmin <= thr_pre) | bnd_any)) {
                RG_TPE_DIAG_REPLAY(it0, j0, CH);
                // rare: replay the chunk with the exact float tests of _validate (roboEnv.py:82-94)
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    ox[i] = ox0[i];
                    oy[i] = oy0[i];
                    x[i] = bx[i] + ox[i];
                    y[i] = by[i] + oy[i];
                    c[i] = c0[i];
                    s[i] = s0[i];
                }
                for (int u = 0; u < C; ++u) {
                    const int code = validate(x, y, c, s);
                    advance(x, y, ox, oy, c, s);  // the violating sub-step is still integrated
                    if (code) {
                        viol = code;
                        n_exec = j0 + u + 1;
                        return false;
                    }
                }
            }
            return true;
        };
        int j = 0;
        bool ok = true;
        for (; ok && j + CH <= n; j += CH) ok = run_chunk(std::integral_constant<int, CH>{}, j);
        sfor<1, CH>([&](auto RR) {  // the remainder as one shorter chunk
            if (ok && n - j == decltype(RR)::value) ok = run_chunk(RR, j);
        });
        // period end: heading and distance for the sub-steps this env executed
        const float ne = static_cast<float>(n_exec);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float adv = __builtin_fabsf(dtv[i]);
            th[i] = wrap_spec(__builtin_fmaf(ne, dtw[i], th[i]));
            acc[i] = __builtin_fmaf(ne, adv, acc[i]);
            last[i] = adv;
            // base <- base + displacement (= x, y as last formed); displacement <- the exact remainder (TwoSum)
            const float tx = x[i] - bx[i], ty = y[i] - by[i];
            ox[i] = (bx[i] - (x[i] - tx)) + (ox[i] - tx);
            oy[i] = (by[i] - (y[i] - ty)) + (oy[i] - ty);
            bx[i] = x[i];
            by[i] = y[i];
        }
        RG_PIN_ARR3(N, x, y, th);
        RG_TSTAMP_PERIOD(it0);
    }
    RG_STAMP(3);  // all periods
    float dist[N];
#pragma unroll
    for (int i = 0; i < N; ++i) dist[i] = viol ? acc[i] : acc[i] - last[i];

    // ---- scenario epilogue
    const int D = p.obs_dim;
    bool done = false;
    int remaining = -1;
    float reward[N];
    const int steps = q_steps[e] + 1;
    if (a.auto_reset) rc_raw = a.st.reset_count[e];  // for the fused reset, should this env finish
    // the statistics words ride the same memory round trip (the pair arrays of the QP are dead here: no pressure)
    float st_ret = 0.0f, st_sum = 0.0f;
    int st_cnt = 0, st_steps = 0;
    if (stats) {
        st_ret = q_ret[e];
        st_sum = q_sum[e];
        st_cnt = q_cnt[e];
        st_steps = q_stp[e];
    }

    if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) {
        const int P = p.num_prey;
        const float *pl = a.st.prey_loc + static_cast<size_t>(e) * 2 * P;
        uint8_t *sen = a.st.prey_sensed + static_cast<size_t>(e) * P, *cap = a.st.prey_captured + static_cast<size_t>(e) * P;
        float sr2[N], cr2[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            sr2[i] = p.sensing_radius[i] * p.sensing_radius[i];
            cr2[i] = p.capture_radius[i] * p.capture_radius[i];
        }
        int unseen0 = 0, left0 = 0, unseen1 = 0, left1 = 0;
        float closest[N], qx[N], qy[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            closest[i] = -1.0f;
            qx[i] = -5.0f;
            qy[i] = -5.0f;
        }
        // up to 8 prey (the reference's configurations: 6): the whole block and its flags are fetched before the
        // loop -- one memory round trip instead of one per prey
        float2 pre_xy[8];
        int pre_s[8], pre_c[8];
        const bool pre = P <= 8;
        if (pre) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int tt = t < P ? t : P - 1;
                pre_xy[t] = reinterpret_cast<const float2 *>(pl)[tt];
                pre_s[t] = sen[tt];
                pre_c[t] = cap[tt];
            }
        }
        auto prey_step = [&](float px, float py, bool sensed, bool captured, int q) {
            unseen0 += !sensed;
            left0 += !captured;
            float d2[N];
            bool any_s = false, any_c = false;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const float dx = x[i] - px, dy = y[i] - py;
                d2[i] = dx * dx + dy * dy;
                any_s = any_s | (d2[i] <= sr2[i]);
                any_c = any_c | ((act[i] == 4) & (d2[i] <= cr2[i]));
            }
            if (!captured) {
                if (!sensed && any_s) sensed = true;
                if (sensed && any_c) captured = true;
            }
            sen[q] = sensed;
            cap[q] = captured;
            unseen1 += !sensed;
            left1 += !captured;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool take = !captured & (d2[i] <= sr2[i]) & ((d2[i] < closest[i]) | (closest[i] == -1.0f));
                qx[i] = take ? px : qx[i];
                qy[i] = take ? py : qy[i];
                closest[i] = take ? d2[i] : closest[i];
            }
        };
        if (pre) {  // a11 tracking + a13 nearest prey, prey by prey (PredatorCapturePrey.py:72-95)
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < P) prey_step(pre_xy[t].x, pre_xy[t].y, pre_s[t] != 0, pre_c[t] != 0, t);
        } else {
            for (int q = 0; q < P; ++q) prey_step(pl[2 * q], pl[2 * q + 1], sen[q] != 0, cap[q] != 0, q);
        }
        RG_PIN_ARR2(N, qx, qy);
        RG_TSTAMP_EPI(0);  // prey tracked
        if (p.capability_aware) {
            float own[N][6];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                own[i][0] = x[i];
                own[i][1] = y[i];
                own[i][2] = qx[i];
                own[i][3] = qy[i];
                own[i][4] = p.sensing_radius[i];
                own[i][5] = p.capture_radius[i];
            }
            write_obs_staged<N, 6, 6 * N>(x, y, own, p.num_neighbors, sg, sv.io.obs, D, [](float *) {});
        } else {
            float own[N][4];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                own[i][0] = x[i];
                own[i][1] = y[i];
                own[i][2] = qx[i];
                own[i][3] = qy[i];
            }
            write_obs_staged<N, 4, 4 * N>(x, y, own, p.num_neighbors, sg, sv.io.obs, D, [](float *) {});
            RG_TSTAMP_EPI(1);  // observation rows written and copied out
        }
        float r;
        if (viol) {
            r = p.violation_reward;
            done = true;
        } else {
            r = 0.0f;
            r = r + static_cast<float>(unseen0 - unseen1) * p.sense_reward;
            r = r + static_cast<float>(left0 - left1) * p.capture_reward;
            r = r + p.time_penalty;
            if (steps > p.max_episode_steps || left1 == 0) {
                done = true;
                remaining = left1;
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) reward[i] = r;
    } else if constexpr (SCN == RG_SCN_WAREHOUSE) {
        uint8_t loaded[N];
        float own[N][3];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            loaded[i] = a.st.loaded[eN + i];
            own[i][0] = x[i];
            own[i][1] = y[i];
            own[i][2] = loaded[i] ? 1.0f : 0.0f;
        }
        write_obs_staged<N, 3, 3 * N>(x, y, own, p.num_neighbors, sg, sv.io.obs, D, [](float *) {});
        if (viol) {
#pragma unroll
            for (int i = 0; i < N; ++i) reward[i] = p.violation_reward;
            done = true;
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool green = (i % 2) == 0;
                float r = 0.0f;
                if (loaded[i]) {
                    if (x[i] < -1.5f + p.goal_width && ((green && y[i] > 0.0f) || (!green && y[i] <= 0.0f))) {
                        r = p.unload_reward;
                        loaded[i] = 0;
                    }
                } else {
                    if (x[i] > 1.5f - p.goal_width && ((!green && y[i] > 0.0f) || (green && y[i] <= 0.0f))) {
                        r = p.load_reward;
                        loaded[i] = 1;
                    }
                }
                reward[i] = r;
                a.st.loaded[eN + i] = loaded[i];
            }
            done = steps > p.max_episode_steps;
        }
    } else if constexpr (SCN == RG_SCN_SIMPLE) {
        const float goal_x = a.st.prey_loc[static_cast<size_t>(e) * 2], goal_y = a.st.prey_loc[static_cast<size_t>(e) * 2 + 1];
        float own[N][2];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            own[i][0] = x[i];
            own[i][1] = y[i];
        }
        write_obs_staged<N, 2, 2 * N + 2>(x, y, own, N - 1, sg, sv.io.obs, D, [&](float *row) {  // simple.py:98: D = 2 (N + 1)
            row[2 * N] = goal_x;
            row[2 * N + 1] = goal_y;
        });
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (viol) {
                reward[i] = p.violation_reward;
            } else {
                const float dx = x[i] - goal_x, dy = y[i] - goal_y;
                const float r = -(dx * dx + dy * dy);
                reward[i] = r * p.reward_scaler;
            }
        }
        done = viol ? true : steps > p.max_episode_steps;
    } else if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) {
        static_assert(SCN != RG_SCN_ARCTIC_TRANSPORT || N == 4, "ArcticTransport has 4 agents");
        const uint8_t *grid = a.st.grid + static_cast<size_t>(e) * 96;
        const int gc = a.st.goal_col[e];
        int row[N], col[N], reached[N], here[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            int r_ = -static_cast<int>((y[i] - 1.0f) / 0.25f), c_ = static_cast<int>((x[i] + 1.5f) / 0.25f);
            row[i] = r_ < 0 ? 0 : r_ > 7 ? 7 : r_;
            col[i] = c_ < 0 ? 0 : c_ > 11 ? 11 : c_;
            here[i] = grid[row[i] * 12 + col[i]];
            pix[i] = here[i];
            reached[i] = a.st.reached_goal[eN + i] | (here[i] == 3 ? 1 : 0);
        }
        const float goalx = static_cast<float>(gc) * 0.25f - 1.5f, goaly = -1.0f * 0.25f + 0.75f;
        float surround[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int left = col[i] > 0 ? col[i] - 1 : col[i], right = col[i] < 11 ? col[i] + 1 : col[i];
            const int up = row[i] > 0 ? row[i] - 1 : row[i], down = row[i] < 7 ? row[i] + 1 : row[i];
            surround[i][0] = static_cast<float>(grid[up * 12 + left]);
            surround[i][1] = static_cast<float>(grid[row[i] * 12 + left]);
            surround[i][2] = static_cast<float>(grid[down * 12 + left]);
            surround[i][3] = static_cast<float>(grid[up * 12 + col[i]]);
            surround[i][4] = static_cast<float>(grid[down * 12 + col[i]]);
            surround[i][5] = static_cast<float>(grid[up * 12 + right]);
            surround[i][6] = static_cast<float>(grid[row[i] * 12 + right]);
            surround[i][7] = static_cast<float>(grid[down * 12 + right]);
        }
        stage_obs_rows<N, 30>(sg, sv.io.obs, D, [&](auto AA, float *o) {  // ArcticTransport.py:19: D = 30
            constexpr int A = decltype(AA)::value;
            constexpr int o0 = A == 0 ? 1 : A == 1 ? 0 : A == 2 ? 3 : 2, o1 = A < 2 ? 2 : 0, o2 = A < 2 ? 3 : 1;
            o[0] = x[A];
            o[1] = y[A];
            o[2] = static_cast<float>(here[A]);
            constexpr int oth[3] = {o0, o1, o2};
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                o[3 + 3 * m + 0] = x[oth[m]];
                o[3 + 3 * m + 1] = y[oth[m]];
                o[3 + 3 * m + 2] = static_cast<float>(here[oth[m]]);
            }
            o[12] = goalx;
            o[13] = goaly;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 8; ++t) o[14 + 8 * i + t] = surround[i][t];
        });
        float r;
        if (viol) {
            r = p.violation_reward;
            done = true;
        } else {
            r = 0.0f;
#pragma unroll
            for (int jj = 2; jj < 4; ++jj) {
                if (!reached[jj]) r = r + p.not_reached_penalty;
                if (pix[jj] != 3) {
                    const float dx = x[jj] - goalx, dy = y[jj] - goaly;
                    r = r + p.dist_multiplier * (dx * dx + dy * dy);
                }
            }
            done = steps > p.max_episode_steps;
            if (!done) done = reached[2] && reached[3];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            reward[i] = r;
            a.st.pixel_type[eN + i] = static_cast<uint8_t>(pix[i]);
            a.st.reached_goal[eN + i] = static_cast<uint8_t>(reached[i]);
        }
    } else {  // MaterialTransport
        int msg[4], load[N];
#pragma unroll
        for (int i = 0; i < 4; ++i) msg[i] = i < N ? act[i < N ? i : 0] % 4 : a.st.messages[4 * e + i];
        int zone0 = a.st.zone_load[2 * e], zone1 = a.st.zone_load[2 * e + 1];
#pragma unroll
        for (int i = 0; i < N; ++i) load[i] = a.st.load[eN + i];
        stage_obs_rows<N, 11>(sg, sv.io.obs, D, [&](auto II, float *o) {  // D = 9, or 11 with the capabilities
            constexpr int i = decltype(II)::value;
            o[0] = x[i];
            o[1] = y[i];
            o[2] = static_cast<float>(load[i]);
            o[3] = static_cast<float>(zone0);
            o[4] = static_cast<float>(zone1);
#pragma unroll
            for (int t = 0; t < 4; ++t) o[5 + t] = static_cast<float>(msg[t]);
            if (p.capability_aware) {
                o[9] = static_cast<float>(p.torque[i]);
                o[10] = p.agent_step[i];
            }
        });
        float r;
        if (viol) {
            r = p.violation_reward;
            done = true;
        } else {
            r = p.time_penalty;
            const float egw = p.end_goal_width, zr2 = p.zone1_radius * p.zone1_radius;
            bool any_load = false;
#pragma unroll
            for (int i = 0; i < N; ++i) {  // sequential over agents (MaterialTransport.py:161-189)
                const int tq = p.torque[i];
                if (load[i] > 0) {
                    if (x[i] < -1.5f + egw) {
                        r = r + static_cast<float>(load[i]) * p.unload_multiplier;
                        load[i] = 0;
                    }
                } else {
                    if (x[i] > 1.5f - egw) {
                        if (zone1 > tq) {
                            load[i] = tq;
                            zone1 -= tq;
                        } else {
                            load[i] = zone1;
                            zone1 = 0;
                        }
                        r = r + static_cast<float>(load[i]) * p.load_multiplier;
                    } else if (x[i] * x[i] + y[i] * y[i] <= zr2) {
                        if (zone0 > tq) {
                            load[i] = tq;
                            zone0 -= tq;
                        } else {
                            load[i] = zone0;
                            zone0 = 0;
                        }
                        r = r + static_cast<float>(load[i]) * p.load_multiplier;
                    }
                }
                any_load = any_load || (load[i] != 0);
            }
            done = steps > p.max_episode_steps;
            if (!done) done = (zone0 == 0 && zone1 == 0 && !any_load);
        }
        if (done) {
            remaining = zone0 + zone1;
#pragma unroll
            for (int i = 0; i < N; ++i) remaining += load[i];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            reward[i] = r;
            a.st.load[eN + i] = load[i];
        }
        a.st.zone_load[2 * e] = zone0;
        a.st.zone_load[2 * e + 1] = zone1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < N) a.st.messages[4 * e + i] = msg[i];
    }

    // ---- stores
    RG_STAMP(4);  // scenario epilogue (observation rows already on their way)
    bool trunc = false;
    {
        float rsum = 0.0f;
        {   // poses, carried distance, rewards and distances: the wave's span of each array through the LDS block
            static_assert(WAVE * 6 * N <= stage_dw<N>(), "the small records of the wave must fit the staging block");
            float *sp = sg.buf + sg.lane * (3 * N), *sc = sg.buf + WAVE * (3 * N) + sg.lane * N;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                sp[i] = x[i];
                sp[N + i] = y[i];
                sp[2 * N + i] = th[i];
                sc[i] = last[i];
                sc[WAVE * N + i] = reward[i];
                sc[2 * WAVE * N + i] = dist[i];
                rsum = rsum + reward[i];
            }
            stage_fence();
            const size_t w0 = sg.env0 * N;
            const size_t EN = static_cast<size_t>(a.E) * N;   // (array bounds: read by -DRG_TPE_GUARD builds only)
            copy_span<3 * N>(sg, sg.buf, a.st.poses + w0 * 3, sg.nact * (3 * N), a.st.poses, a.st.poses + EN * 3);
            copy_span<N>(sg, sg.buf + WAVE * (3 * N), a.st.carry_dist + w0, sg.nact * N, a.st.carry_dist, a.st.carry_dist + EN);
            copy_span<N>(sg, sg.buf + WAVE * (4 * N), sv.io.reward + w0, sg.nact * N, sv.io.reward, sv.io.reward + EN);
            copy_span<N>(sg, sg.buf + WAVE * (5 * N), sv.io.dist_travelled + w0, sg.nact * N, sv.io.dist_travelled, sv.io.dist_travelled + EN);
            stage_fence();
        }
        a.st.episode_steps[e] = steps;
        // gym's TimeLimit on top of the scenario (gymma block of rg_step_io)
        // (the counter is fetched here, not with the state: this kernel has no register to spare -- one more value live
        // through the step costs its second wave per SIMD)
        if (int32_t *q_el = sv.io.elapsed) {
            const int el_raw = q_el[e];
            trunc = !done & (el_raw + 1 >= sv.io.time_limit);
            q_el[e] = (done | trunc) ? 0 : el_raw + 1;
            sv.io.truncated[e] = trunc ? 1 : 0;
            sv.io.ended[e] = (done | trunc) ? 1 : 0;
            sv.io.reward_sum[e] = rsum;
            // rg_step_io.zero_obs_on_end: an env that ends hands the trainer the reset observation (zeros).  Its rows were
            // stored above (by this wave: staged through LDS or written by this lane); this wave's later stores win.
            if (sv.io.zero_obs_on_end && (done | trunc)) {
                float *rows = sv.io.obs + sg.e * N * p.obs_dim;
                for (int c = 0; c < N * p.obs_dim; ++c) rows[c] = 0.0f;
            }
        }
        if (stats) {  // misc.py:178-185
            float ret = st_ret + (p.shared_reward ? reward[0] : rsum);
            if (done | trunc) {  // a truncated episode counts like a finished one
                a.st.done_return_sum[e] = st_sum + ret;
                a.st.done_count[e] = st_cnt + 1;
                a.st.done_steps_sum[e] = st_steps + steps;
                ret = 0.0f;
            }
            a.st.ep_return[e] = ret;
        }
        sv.io.done[e] = done ? 1 : 0;
        sv.io.violation[e] = static_cast<uint8_t>(viol);
        sv.io.remaining[e] = remaining;
        RG_TPE_DIAG_REPORT(max_sweeps);
        if (sv.io.qp_sweeps) sv.io.qp_sweeps[e] = max_sweeps;
    }
    RG_STAMP_ALWAYS_5();
    RG_STAMPS_WRITE(sg.lane, sv.io.qp_sweeps, e, a.E, max_sweeps)
    return done | trunc;
}

// ------------------------------------------------------------------ the step kernel
// QPM = RG_QP_CVXOPT: the interior-point mode's own instantiations, compiled for one wave per SIMD (the iteration of ipm_qp.h
// wants the whole register file; the launch is then ~90 % that iteration).
template <int SCN, int N, bool ROLLOUT, int QPM = 0>
// Waves per SIMD (tpe_waves above; these files are compiled without the SLP vectoriser, build.py FILE_FLAGS, which alone
// took N = 5 from 244 to 203 VGPRs, N <= 4 from 168 + 7 spilled to 147 and N = 6 from 76 to 19 spilled values): N <= 4
// three, N = 5 two -- both now the compiler's own allocation --, N = 6 two on a forced 256-register budget (19 values in
// scratch; one wave per SIMD measured 34 % slower in round 2), N >= 7 one.  Measured and rejected in round 3 at 524 288 envs
// (tools/tpe_ab_probe.py, -DRG_TPE_W5=3 / -DRG_TPE_W4=4 / -DRG_TPE_W78=2): N = 5 at three waves (32 spilled) 169.6 vs
// 153.1 us, N = 4 at four (18 spilled) 108.7 vs 99.5, N = 7 at two (93 spilled) 390.6 vs 348.0, N = 8 at two 1448 vs 481.
RG_TPE_WAVES_ATTR(QPM ? 1 : tpe_waves(N) ? tpe_waves(N) : 1)
__global__ __launch_bounds__(WAVE) void step_kernel(const KernelArgs a) {
    __shared__ union alignas(16) {
        Lds<WAVE> reset;        // fused reset (after the step, behind a barrier)
        float stage[stage_dw<N>()];  // the step's stores
    } shm;
    Lds<WAVE> &lds = shm.reset;
    const int chunk = xcd_chunk();
    // All 64 lanes run the step (the staged stores are copied out by the whole wave): lanes past the end of the batch
    // repeat its last env -- the same loads, the same values, the same stores -- and take no part in the fused reset.
    const int e_raw = chunk * WAVE + threadIdx.x;
    // host simulation (tests/sanitize/): lanes are threads, not in lock step, so the surplus lanes of the last wave cannot share
    // env E - 1's read-modify-writes; the harness pads every array to whole waves with copies of that env
    const int e = (kHostSim || e_raw < a.E) ? e_raw : a.E - 1;
    const int left = a.E - chunk * WAVE;
    const Stage sg{shm.stage, static_cast<int>(threadIdx.x), left < WAVE ? left : WAVE, static_cast<size_t>(chunk) * WAVE,
                   static_cast<size_t>(e), a.st.done_count, a.E};
    const int num_steps = ROLLOUT ? a.num_steps : 1;  // rg_rollout: no device-wide synchronisation between steps
    for (int t = 0; t < num_steps; ++t) {
        if (t) __syncthreads();  // the previous step's stores and resets are visible to the wave
        int rc_raw = -1;
        const bool done = step_env<SCN, N, QPM>(a, step_view(a, t, N, a.p.obs_dim), e, rc_raw, sg);
        // fused auto-reset (scenario.reset(); ~1 env in 70 per step): the whole wave resets each finished
        // env together, as one 64-lane group of the shared sampler
        if (a.auto_reset) {
            unsigned long long todo = __ballot(done & (e_raw < a.E));
            if (todo) __syncthreads();  // the wave's state stores are complete before other lanes rewrite them
            while (todo) {
                const int i = __builtin_ctzll(todo);
                todo &= todo - 1;
                reset_group<SCN, WAVE>(a, lds, chunk * WAVE + i, 0, threadIdx.x, true, __builtin_amdgcn_readlane(rc_raw, i));
            }
        }
    }
}

template <int SCN, bool ROLLOUT>
static hipError_t launch_scn(const KernelArgs &a, hipStream_t stream) {
    const int grid = (a.E + WAVE - 1) / WAVE;
    if (a.p.qp_mode == RG_QP_CVXOPT) {   // N <= 5: the iteration's rows and KKT matrix in one lane's registers (tpe_supported)
        switch (a.p.n_agents) {
#define RG_CASE(NN)                                                                                                      \
    case NN:                                                                                                             \
        hipLaunchKernelGGL((step_kernel<SCN, NN, ROLLOUT, RG_QP_CVXOPT>), dim3(grid), dim3(WAVE), 0, stream, a);         \
        break;
            RG_CASE(2)
            RG_CASE(3)
            RG_CASE(4)
            RG_CASE(5)
#undef RG_CASE
            default:
                return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (a.p.n_agents) {
#define RG_CASE(NN)                                                                                   \
    case NN:                                                                                          \
        hipLaunchKernelGGL((step_kernel<SCN, NN, ROLLOUT>), dim3(grid), dim3(WAVE), 0, stream, a);             \
        break;
        // N <= 6: the agent counts the library dispatches this kernel for.  N = 7, 8 compile (tools/n7_bisect/, tests/sanitize/) but are
        // not part of the library since round 4: never chosen (the lane-group kernel is faster there at every batch size), and the
        // one instantiation the ROCm 7.2 register allocator miscompiled under some flags.
        RG_CASE(2)
        RG_CASE(3)
        RG_CASE(4)
        RG_CASE(5)
        RG_CASE(6)
#undef RG_CASE
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace tpe

template <bool ROLLOUT>
static hipError_t launch_tpe(const KernelArgs &a, hipStream_t stream) {
    switch (a.p.scenario) {
        case RG_SCN_PREDATOR_CAPTURE_PREY:
            return tpe::launch_scn<RG_SCN_PREDATOR_CAPTURE_PREY, ROLLOUT>(a, stream);
        case RG_SCN_WAREHOUSE:
            return tpe::launch_scn<RG_SCN_WAREHOUSE, ROLLOUT>(a, stream);
        case RG_SCN_MATERIAL_TRANSPORT:
            return tpe::launch_scn<RG_SCN_MATERIAL_TRANSPORT, ROLLOUT>(a, stream);
        case RG_SCN_SIMPLE:
            return tpe::launch_scn<RG_SCN_SIMPLE, ROLLOUT>(a, stream);
        case RG_SCN_ARCTIC_TRANSPORT: {
            const int grid = (a.E + WAVE - 1) / WAVE;
            if (a.p.qp_mode == RG_QP_CVXOPT)
                hipLaunchKernelGGL((tpe::step_kernel<RG_SCN_ARCTIC_TRANSPORT, 4, ROLLOUT, RG_QP_CVXOPT>), dim3(grid), dim3(WAVE), 0, stream, a);
            else
                hipLaunchKernelGGL((tpe::step_kernel<RG_SCN_ARCTIC_TRANSPORT, 4, ROLLOUT>), dim3(grid), dim3(WAVE), 0, stream, a);
            return hipGetLastError();
        }
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace rg
