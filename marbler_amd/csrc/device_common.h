// device_common.h -- device code shared by the two step kernels (robogym_kernels.hip: lane group
// per env; robogym_tpe.hip: thread per env): DPP lane exchange, group reductions, the one-wavefront
// LDS scratch and the reset sampler.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/robogym.h"
#include "kernel_args.h"
#include "sim_math.h"

namespace rg {

constexpr int WAVE = 64;
constexpr int MAX_DRAWS = 128;  // u32 draws per reset: 4 + 2N + P <= 4 + 32 + 64, rounded up to blocks

// ------------------------------------------------------------------ workgroup -> env chunk (XCD-aware)
// Workgroups are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8), each with its own L2.  With
// chunk = b, neighbouring env chunks -- whose state and output spans share 128-byte lines wherever a span
// is not a multiple of the line (5 agents: 60-byte pose blocks, 20-byte reward rows, 1-byte done flags) --
// land on different L2s: every shared line is fetched by two XCDs and written back as two partial lines.
// Instead XCD x works through ONE contiguous run of chunks, so shared lines meet in one L2 (measured with
// rocprofv3 FETCH_SIZE / WRITE_SIZE, profiles/).  Affinity only: results do not depend on the block -> XCD map.
__device__ __forceinline__ int xcd_chunk(int G = gridDim.x) {
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int q = G >> 3, r = G & 7;
    return xcd * q + (xcd < r ? xcd : r) + j;
}

// ------------------------------------------------------------------ lane exchange (DPP)
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// value held by lane (lane ^ K), K in 1..15, within a 16-lane row
template <int K>
__device__ __forceinline__ int xor_lane_i(int v) {
    static_assert(K >= 1 && K <= 15, "xor distance");
    constexpr int QP1 = 0xB1, QP2 = 0x4E, QP3 = 0x1B;  // quad_perm [1,0,3,2] [2,3,0,1] [3,2,1,0]
    constexpr int HALF_MIRROR = 0x141, MIRROR = 0x140, ROR8 = 0x128;
    if constexpr (K == 1) return dpp_i<QP1>(v);
    else if constexpr (K == 2) return dpp_i<QP2>(v);
    else if constexpr (K == 3) return dpp_i<QP3>(v);
    else if constexpr (K == 7) return dpp_i<HALF_MIRROR>(v);
    else if constexpr (K == 4) return dpp_i<QP3>(dpp_i<HALF_MIRROR>(v));
    else if constexpr (K == 5) return dpp_i<QP2>(dpp_i<HALF_MIRROR>(v));
    else if constexpr (K == 6) return dpp_i<QP1>(dpp_i<HALF_MIRROR>(v));
    else if constexpr (K == 15) return dpp_i<MIRROR>(v);
    else if constexpr (K == 8) return dpp_i<ROR8>(v);
    else if constexpr (K == 9) return dpp_i<QP1>(dpp_i<ROR8>(v));
    else if constexpr (K == 10) return dpp_i<QP2>(dpp_i<ROR8>(v));
    else if constexpr (K == 11) return dpp_i<QP3>(dpp_i<ROR8>(v));
    else if constexpr (K == 12) return dpp_i<QP3>(dpp_i<MIRROR>(v));
    else if constexpr (K == 13) return dpp_i<QP2>(dpp_i<MIRROR>(v));
    else return dpp_i<QP1>(dpp_i<MIRROR>(v));
}
template <int K>
__device__ __forceinline__ float xor_lane(float v) {
    return __builtin_bit_cast(float, xor_lane_i<K>(__builtin_bit_cast(int, v)));
}

// groups of 8 lanes: lane M of the lower quad's neighbour quad, i.e. lanes 0..3 receive lane 4+M and
// lanes 4..7 receive lane M (quad broadcast, then half-row mirror)
template <int M>
__device__ __forceinline__ int cross_lane_i(int v) {
    static_assert(M >= 0 && M <= 3, "quad lane");
    return dpp_i<0x141>(dpp_i<(M | (M << 2) | (M << 4) | (M << 6))>(v));
}
template <int M>
__device__ __forceinline__ float cross_lane(float v) {
    return __builtin_bit_cast(float, cross_lane_i<M>(__builtin_bit_cast(int, v)));
}

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
// d[i] <- bits of d[i].x * d[i].x + d[i].y * d[i].y in binary32, i < K: VOP3P v_dot2_f32_f16 with a literal-zero addend
// (the compiler's own selection for __builtin_amdgcn_fdot2 is v_dot2c, which costs an extra v_mov to clear the
// accumulator), issued as ONE inline-asm block that ends in `s_nop 2`.
// HAZARD (gfx940 / gfx950): a VALU instruction that reads the destination of a DOT instruction needs 3 wait states
// after it; the hardware does not interlock, and the compiler's hazard recogniser cannot see into inline asm (for its
// own v_dot2c it emits exactly this s_nop 2).  With a bare `asm("v_dot2_f32_f16 ...")` a consumer scheduled within
// three instructions of the dot reads the register's OLD contents -- the packed binary16 difference itself.  Rounds 1-2
// shipped that form and were correct only because the scheduler happened to issue 20 dots back to back ahead of their
// first consumer; the first restructuring of the pre-test (round 3) put a v_min two instructions behind a dot and every
// chunk went to the exact replay.  The block below is safe wherever it is scheduled: later dots of the block are the
// wait states of the earlier ones, the s_nop covers the last.
template <int K>
__device__ __forceinline__ void dot2_batch(int (&d)[K]) {
    static_assert(K >= 1, "at least one");
#ifdef RG_HOST_SIM  // host simulation under the CPU sanitizers (tests/sanitize/): the same dot products in C
    for (int i = 0; i < K; ++i) d[i] = rg_sim_dot2_self(d[i]);
    return;
#endif
    if constexpr (K >= 4) {
        asm("v_dot2_f32_f16 %0, %0, %0, 0\n\tv_dot2_f32_f16 %1, %1, %1, 0\n\tv_dot2_f32_f16 %2, %2, %2, 0\n\t"
            "v_dot2_f32_f16 %3, %3, %3, 0\n\ts_nop 2"
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
        if constexpr (K > 4) {
            int rest[K - 4];
#pragma unroll
            for (int i = 0; i < K - 4; ++i) rest[i] = d[4 + i];
            dot2_batch<K - 4>(rest);
#pragma unroll
            for (int i = 0; i < K - 4; ++i) d[4 + i] = rest[i];
        }
    } else if constexpr (K == 3) {
        asm("v_dot2_f32_f16 %0, %0, %0, 0\n\tv_dot2_f32_f16 %1, %1, %1, 0\n\tv_dot2_f32_f16 %2, %2, %2, 0\n\ts_nop 2"
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]));
    } else if constexpr (K == 2) {
        asm("v_dot2_f32_f16 %0, %0, %0, 0\n\tv_dot2_f32_f16 %1, %1, %1, 0\n\ts_nop 2" : "+v"(d[0]), "+v"(d[1]));
    } else {
        asm("v_dot2_f32_f16 %0, %0, %0, 0\n\ts_nop 2" : "+v"(d[0]));
    }
}

template <int I, int END, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, END>(f);
    }
}

// reductions over the lanes of a group: butterflies whose every stage is ONE DPP permute
// (xor 1, xor 2, then the half-row / row mirrors)
template <int GW>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, xor_lane<1>(v));
    v = fmaxf(v, xor_lane<2>(v));
    if constexpr (GW >= 8) v = fmaxf(v, xor_lane<7>(v));
    if constexpr (GW >= 16) v = fmaxf(v, xor_lane<15>(v));
    return v;
}
// sum over the lanes of a group, the same butterfly: ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)) ... -- every lane ends
// with the same bits (each stage adds the same two partial sums on both sides; binary32 addition commutes)
template <int GW>
__device__ __forceinline__ float group_sum(float v) {
    v = v + xor_lane<1>(v);
    v = v + xor_lane<2>(v);
    if constexpr (GW >= 8) v = v + xor_lane<7>(v);
    if constexpr (GW >= 16) v = v + xor_lane<15>(v);
    return v;
}
// the same for NON-NEGATIVE floats, on their bit patterns (order-preserving; lets the DPP permute
// fuse into v_max_u32 and needs no NaN canonicalisation)
template <int GW>
__device__ __forceinline__ float group_max_nonneg(float f) {
    uint32_t v = __builtin_bit_cast(uint32_t, f);
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, static_cast<uint32_t>(xor_lane_i<1>(static_cast<int>(v))));
    v = mx(v, static_cast<uint32_t>(xor_lane_i<2>(static_cast<int>(v))));
    if constexpr (GW >= 8) v = mx(v, static_cast<uint32_t>(xor_lane_i<7>(static_cast<int>(v))));
    if constexpr (GW >= 16) v = mx(v, static_cast<uint32_t>(xor_lane_i<15>(static_cast<int>(v))));
    return __builtin_bit_cast(float, v);
}
template <int GW>
__device__ __forceinline__ uint32_t group_or(uint32_t v) {
    v |= static_cast<uint32_t>(xor_lane_i<1>(static_cast<int>(v)));
    v |= static_cast<uint32_t>(xor_lane_i<2>(static_cast<int>(v)));
    if constexpr (GW >= 8) v |= static_cast<uint32_t>(xor_lane_i<7>(static_cast<int>(v)));
    if constexpr (GW >= 16) v |= static_cast<uint32_t>(xor_lane_i<15>(static_cast<int>(v)));
    return v;
}
// does any lane of my group have `pred` set?  (rare paths only: one ballot, then bit tests)
template <int GW>
__device__ __forceinline__ bool group_any(bool pred, int gbase) {
    const unsigned long long m = __ballot(pred);
    constexpr unsigned long long GM = (GW == 64) ? ~0ull : ((1ull << GW) - 1ull);
    return ((m >> gbase) & GM) != 0ull;
}

// ------------------------------------------------------------------ LDS scratch (one wavefront)
// The step's scratch and the reset sampler's are never live together (the fused reset runs after the step's last LDS
// read, an explicit reset runs alone), so they share the block: 7.5 KB instead of 13.5 KB per one-wave workgroup --
// at 13.5 KB a CU's 160 KB held 11 workgroups, 2.75 waves per SIMD, whatever the register count allowed.
template <int GW>
struct alignas(16) Lds {
    union {
        struct {  // step
            float prey[WAVE / GW][RG_MAX_PREY * 2];  // the env's prey block (PredatorCapturePrey)
            float own[WAVE][8];                      // each agent's own-observation row (<= 6 floats)
            float ax[WAVE], ay[WAVE];                // MaterialTransport sequential replay / reward sum
            int aload[WAVE];
            uint8_t grid[WAVE / GW][RG_ARCTIC_ROWS * RG_ARCTIC_COLS];  // ArcticTransport terrain of the env
        };
        struct {  // reset
            uint32_t draws[WAVE / GW][MAX_DRAWS];    // Philox output
            uint8_t perm[WAVE / GW][2][64];          // Fisher-Yates permutations of the grid cells (agents, prey)
            uint8_t sel[WAVE / GW][2][64];           // chosen cells
        };
    };
};

// ------------------------------------------------------------------ reset sampler (a17)
// misc.py:49-63 -> rps generate_initial_conditions (Appendix A.7): N distinct cells of an
// nx x ny grid, then the scenario's shift.  The reference draws from NumPy's global MT19937;
// here every (global env, episode) pair owns a Philox4x32-10 stream (DESIGN.md "reset").  Draw
// order: [MaterialTransport: 4 for the two zone loads] N cells, N headings, [PCP: P prey cells].
// The whole wave calls this (wave-uniform); groups with do_reset set take part: Philox blocks
// and the permutation fill are spread over the group's lanes, the short Fisher-Yates chain runs
// on the group's lane 0, and every lane writes its own agent's state.
__device__ __forceinline__ float uniform01(uint32_t r) { return static_cast<float>(r >> 8) * 5.9604644775390625e-08f; }

__device__ __forceinline__ int normal_int(uint32_t r1, uint32_t r2, float mean, float stdv) {
    // int(np.random.normal(mean, std)) by Box-Muller on the spec'd log / sincos
    const float u1 = static_cast<float>((r1 >> 8) + 1u) * 5.9604644775390625e-08f;  // (0, 1]
    const float u2 = uniform01(r2);
    const float rad = __builtin_sqrtf(-2.0f * log_spec(u1));
    float sn, cs;
    sincos_spec(u2 * 6.283185482025146484375f - 3.1415927410125732421875f, sn, cs);
    return static_cast<int>(mean + stdv * (rad * cs));
}

// Partial Fisher-Yates over the grid cells.  The agents' draw and (PredatorCapturePrey) the prey's
// draw are independent chains, so they run side by side on lanes 0 and 1 of the group, each on its
// own permutation array (same instructions, different data).
template <int GW>
__device__ __forceinline__ void fisher_yates2(Lds<GW> &lds, int g, int ag, bool do_reset, const rg_grid &grid_a,
                                              int count_a, int first_a, const rg_grid &grid_b, int count_b,
                                              int first_b) {
    const int Ca = grid_a.nx * grid_a.ny, Cb = count_b > 0 ? grid_b.nx * grid_b.ny : 0;
    constexpr int RMAX = 8;
    if (count_a <= RMAX && count_b <= RMAX) {
        // short draws (the reference's scenarios: 4..8 agents, 6 prey): the same partial Fisher-Yates without
        // the permutation array.  Position p holds p unless an earlier step k swapped something into it
        // (j_k == p, value t_k = what position k held then); the latest such step wins.  All in registers:
        // no dependent LDS round trips on the chain.
        __syncthreads();  // the Philox draws are in LDS
        if (do_reset && ag < 2) {
            const int which = ag;
            const int C = which ? Cb : Ca, count = which ? count_b : count_a, first = which ? first_b : first_a;
            uint32_t r[RMAX];
#pragma unroll
            for (int i = 0; i < RMAX; ++i) r[i] = lds.draws[g][first + (i < count ? i : 0)];
            int jpos[RMAX], tval[RMAX];
            const int cmax = count_a > count_b ? count_a : count_b;
#pragma unroll
            for (int i = 0; i < RMAX; ++i) {
                if (i >= cmax) break;  // wave-uniform
                const int j = i + static_cast<int>((static_cast<uint64_t>(r[i]) * static_cast<uint32_t>(C - i)) >> 32);
                int t = i, sv = j;
#pragma unroll
                for (int q = 0; q < i; ++q) {
                    t = (jpos[q] == i) ? tval[q] : t;
                    sv = (jpos[q] == j) ? tval[q] : sv;
                }
                sv = (j == i) ? t : sv;
                jpos[i] = j;
                tval[i] = t;
                if (i < count) lds.sel[g][which][i] = static_cast<uint8_t>(sv);
            }
        }
        __syncthreads();
        return;
    }
    if (do_reset) {
        for (int i = ag; i < Ca; i += GW) lds.perm[g][0][i] = static_cast<uint8_t>(i);
        for (int i = ag; i < Cb; i += GW) lds.perm[g][1][i] = static_cast<uint8_t>(i);
    }
    __syncthreads();
    if (do_reset && ag < 2) {
        const int which = ag;
        const int C = which ? Cb : Ca, count = which ? count_b : count_a, first = which ? first_b : first_a;
        for (int i = 0; i < count; ++i) {
            const uint32_t r = lds.draws[g][first + i];
            const int j = i + static_cast<int>((static_cast<uint64_t>(r) * static_cast<uint32_t>(C - i)) >> 32);
            const uint8_t t = lds.perm[g][which][i];
            const uint8_t pj = lds.perm[g][which][j];
            lds.perm[g][which][j] = t;
            lds.perm[g][which][i] = pj;
            lds.sel[g][which][i] = pj;
        }
    }
    __syncthreads();
}

// rps: `choices = np.random.choice(x_range * y_range, N, replace=False) + 1`, then `x, y = divmod(c, y_range)`: the
// sampled index is shifted by one before it is split, so cell (0, 0) is never used and (nx, 0) -- one column past the
// grid -- is (recalled from upstream rps/utilities/misc.py; unpinned)
__device__ __forceinline__ void cell_xy(const rg_grid &grid, int cell, float &x, float &y) {
    cell += 1;
    const int cx = cell / grid.ny, cy = cell - cx * grid.ny;
    const float fx = static_cast<float>(cx) * grid.spacing - grid.w2;
    const float fy = static_cast<float>(cy) * grid.spacing - grid.h2;
    x = (fx + grid.ox1) + grid.ox2;
    y = (fy + grid.oy1) + grid.oy2;
}

// Where a draw goes.  commit = true: the env's state arrays, and the episode really starts (flags and counters cleared,
// reset_count advanced).  commit = false: the env's block of rg_state.next_init -- the lane-group step kernel draws an
// env's NEXT initial state ahead of time, at the end of a launch in which the env did not finish, so that the launch in
// which it does finish (typically its slowest wavefront: a near-collision env) only copies it (step_group.h).
struct ResetDst {
    float *poses;     // [3][N]
    float *prey;      // [P][2]   PredatorCapturePrey, Simple
    int32_t *zone;    // [2]      MaterialTransport
    uint8_t *grid;    // [96]     ArcticTransport
    int32_t *gcol;    //          ArcticTransport
    bool commit;
};
__device__ __forceinline__ ResetDst reset_dst_state(const KernelArgs &a, int e) {
    const int N = a.p.n_agents, P = a.p.num_prey;
    ResetDst d;
    d.poses = a.st.poses + static_cast<size_t>(e) * 3 * N;
    d.prey = a.st.prey_loc ? a.st.prey_loc + static_cast<size_t>(e) * 2 * P : nullptr;
    d.zone = a.st.zone_load ? a.st.zone_load + 2 * static_cast<size_t>(e) : nullptr;
    d.grid = a.st.grid ? a.st.grid + static_cast<size_t>(e) * 96 : nullptr;
    d.gcol = a.st.goal_col ? a.st.goal_col + e : nullptr;
    d.commit = true;
    return d;
}
// layout of one env's block of next_init (next_stride floats): poses [3N] | prey [2P] or zone loads [2] (as ints) or
// terrain [24 dwords] + goal column
__device__ __forceinline__ ResetDst reset_dst_next(const KernelArgs &a, int e) {
    const int N = a.p.n_agents;
    float *base = a.st.next_init + static_cast<size_t>(e) * a.next_stride;
    ResetDst d;
    d.poses = base;
    d.prey = base + 3 * N;
    d.zone = reinterpret_cast<int32_t *>(base + 3 * N);
    d.grid = reinterpret_cast<uint8_t *>(base + 3 * N);
    d.gcol = reinterpret_cast<int32_t *>(base + 3 * N + 24);
    d.commit = false;
    return d;
}

// episode_pre >= 0: the env's reset_count, already fetched by the caller (the step kernels prefetch it with the rest of
// the state, so a finished env's reset does not start with a memory round trip of its own)
template <int SCN, int GW>
__device__ __forceinline__ void reset_group(const KernelArgs &a, Lds<GW> &lds, int e, int g, int ag, bool do_reset,
                                            int episode_pre, const ResetDst &dst) {
    const rg_scenario_params &p = a.p;
    const int N = p.n_agents;
    if constexpr (SCN == RG_SCN_ARCTIC_TRANSPORT) {
        // ArcticTransport.py:56-82: fixed start poses; terrain grid uniform in {0,1,2} (draws 0..95,
        // row-major), goal column uniform in 1..11 (draw 96), goal block 2x2 in rows 0-1, row 7
        // columns 1..10 cleared.  (The reference draws the column from Python's `random`.)
        constexpr int CELLS = RG_ARCTIC_ROWS * RG_ARCTIC_COLS;
        int32_t episode = 0;
        if (do_reset) {
            episode = episode_pre >= 0 ? episode_pre : a.st.reset_count[e];
            const uint64_t ge = static_cast<uint64_t>(a.env_offset + e);
            for (int b = ag; 4 * b < CELLS + 1; b += GW) {
                uint32_t blk[4];
                philox4x32_10(static_cast<uint32_t>(ge), static_cast<uint32_t>(ge >> 32),
                              static_cast<uint32_t>(episode), static_cast<uint32_t>(b),
                              static_cast<uint32_t>(a.seed), static_cast<uint32_t>(a.seed >> 32), blk);
#pragma unroll
                for (int t = 0; t < 4; ++t) lds.draws[g][4 * b + t] = blk[t];
            }
        }
        __syncthreads();
        if (do_reset) {
            const int gc = 1 + static_cast<int>((static_cast<uint64_t>(lds.draws[g][CELLS]) * 11u) >> 32);
            for (int i = ag; i < CELLS; i += GW) {
                const int row = i / RG_ARCTIC_COLS, col = i - row * RG_ARCTIC_COLS;
                int val = static_cast<int>((static_cast<uint64_t>(lds.draws[g][i]) * 3u) >> 32);
                if (row <= 1 && (col == gc || col == gc - 1)) val = 3;
                if (row == 7 && col >= 1 && col <= 10) val = 0;
                dst.grid[i] = static_cast<uint8_t>(val);
            }
            if (ag < N) {
                float *X = dst.poses;
                X[ag] = ag == 0 ? -0.3f : ag == 1 ? 0.3f : ag == 2 ? -0.9f : 0.9f;  // ArcticTransport.py:30-33
                X[N + ag] = -0.8f;
                X[2 * N + ag] = 1.57079637050628662109375f;
                if (dst.commit) {
                    a.st.carry_dist[static_cast<size_t>(e) * N + ag] = 0.0f;
                    a.st.pixel_type[static_cast<size_t>(e) * N + ag] = 0;
                    a.st.reached_goal[static_cast<size_t>(e) * N + ag] = 0;
                }
            }
            if (ag == 0) {
                *dst.gcol = gc;
                if (dst.commit) {
                    a.st.reset_count[e] = episode + 1;
                    a.st.episode_steps[e] = 0;
                }
            }
        }
        return;
    }
    constexpr bool HAS_PREY = (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) || (SCN == RG_SCN_SIMPLE);
    const int P = HAS_PREY ? p.num_prey : 0;  // Simple: its single goal is drawn like one prey
    constexpr int ZD = (SCN == RG_SCN_MATERIAL_TRANSPORT) ? 4 : 0;
    const int ndraws = ZD + 2 * N + P;
    int32_t episode = 0;
    if (do_reset) {
        episode = episode_pre >= 0 ? episode_pre : a.st.reset_count[e];
        const uint64_t ge = static_cast<uint64_t>(a.env_offset + e);
        for (int b = ag; 4 * b < ndraws; b += GW) {
            uint32_t blk[4];
            philox4x32_10(static_cast<uint32_t>(ge), static_cast<uint32_t>(ge >> 32), static_cast<uint32_t>(episode),
                          static_cast<uint32_t>(b), static_cast<uint32_t>(a.seed), static_cast<uint32_t>(a.seed >> 32),
                          blk);
            lds.draws[g][4 * b + 0] = blk[0];
            lds.draws[g][4 * b + 1] = blk[1];
            lds.draws[g][4 * b + 2] = blk[2];
            lds.draws[g][4 * b + 3] = blk[3];
        }
    }
    fisher_yates2<GW>(lds, g, ag, do_reset, p.agent_grid, N, ZD, p.prey_grid, P, ZD + 2 * N);  // syncs inside
    if (do_reset && ag < N) {
        float x, y;
        cell_xy(p.agent_grid, lds.sel[g][0][ag], x, y);
        const float th = uniform01(lds.draws[g][ZD + N + ag]) * 6.283185482025146484375f - 3.1415927410125732421875f;
        float *X = dst.poses;
        X[ag] = x;
        X[N + ag] = y;
        X[2 * N + ag] = p.keep_theta ? th : 0.0f;
        if (dst.commit) {
            a.st.carry_dist[static_cast<size_t>(e) * N + ag] = 0.0f;
            if constexpr (SCN == RG_SCN_WAREHOUSE) a.st.loaded[static_cast<size_t>(e) * N + ag] = 0;
            if constexpr (SCN == RG_SCN_MATERIAL_TRANSPORT) {
                a.st.load[static_cast<size_t>(e) * N + ag] = 0;
                if (ag < 4) a.st.messages[4 * e + ag] = 0;
            }
        }
    }
    if constexpr (SCN == RG_SCN_MATERIAL_TRANSPORT) {  // MaterialTransport.py:99-100
        if (do_reset && ag < 2) {
            const float mean = ag == 0 ? p.zone1_mean : p.zone2_mean, sd = ag == 0 ? p.zone1_std : p.zone2_std;
            dst.zone[ag] = normal_int(lds.draws[g][2 * ag], lds.draws[g][2 * ag + 1], mean, sd);
        }
    }
    if constexpr (HAS_PREY) {
        if (do_reset) {
            for (int i = ag; i < P; i += GW) {
                float x, y;
                cell_xy(p.prey_grid, lds.sel[g][1][i], x, y);
                float *pl = dst.prey + 2 * i;
                pl[0] = x;
                pl[1] = y;
                if constexpr (SCN == RG_SCN_PREDATOR_CAPTURE_PREY) {
                    if (dst.commit) {
                        a.st.prey_sensed[static_cast<size_t>(e) * P + i] = 0;
                        a.st.prey_captured[static_cast<size_t>(e) * P + i] = 0;
                    }
                }
            }
        }
    }
    if (do_reset && ag == 0 && dst.commit) {
        a.st.reset_count[e] = episode + 1;
        a.st.episode_steps[e] = 0;
    }
}

// the draw straight into the env's state: scenario.reset()
template <int SCN, int GW>
__device__ __forceinline__ void reset_group(const KernelArgs &a, Lds<GW> &lds, int e, int g, int ag, bool do_reset,
                                            int episode_pre = -1) {
    reset_group<SCN, GW>(a, lds, e, g, ag, do_reset, episode_pre, reset_dst_state(a, e));
}

}  // namespace rg
