// actor_mfma.hip -- the EPyMARL recurrent actor (utilities/rnn_agent.py:5-29: fc1 -> ReLU -> GRUCell ->
// fc2, or rnn_ns_agent.py:5-36 with one such network per agent) for all E x N agents of a batch in
// ONE launch: policy inference for on-device evaluation rollouts.
//
// This is the one dense contraction on the path, so it runs on the matrix cores: f32-input MFMA
// (v_mfma_f32_32x32x2_f32, exact f32 products and sums -- the parity bar against the reference's
// float32 modules is 1e-5, which rules the 16x faster bf16 forms out).  One wavefront owns a tile
// of 32 agent rows and carries it through the whole network:
//     X [32 x I]  --fc1-->  Y1 [32 x H]  --GRU (6 gate tiles per 32 hidden columns)-->  h' [32 x H]  --fc2--> q [32 x A]
// A operands (activations) are read from LDS as float4 = four k-steps; B operands (weights) stream
// from L2 as float4 per lane (each weight matrix is read once per tile; all tiles share it in L2; fc1's
// small ragged matrix is staged in LDS, padded).  The k index of lane half h runs over
// [h K/2, (h+1) K/2): any pairing of k values into MFMA steps gives the same sum up to rounding, and
// this one makes both operands contiguous.  Layer outputs come out of the MFMA with the column on the
// lane and 16 rows in registers (C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5));
// the gate arithmetic is elementwise in that layout, and one LDS write turns it into the next layer's
// row-major A image.  Non-shared actors: a tile takes the rows of ONE agent index (stride N), so the
// whole tile uses one weight set.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/robogym.h"

namespace rg {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int TM = 32;        // agent rows per wavefront
constexpr int MAX_IP = 64;    // padded input width (multiple of 8)

struct ActorArgs {
    rg_actor_weights w;
    const float *obs;       // [E][N][D]
    const uint8_t *restart; // [E] or NULL: nonzero = a new episode: hidden state and observation are taken as zero
    float *hidden;          // [E][N][H] in/out
    float *q;               // [E][N][A] or NULL
    int32_t *actions;       // [E][N] or NULL
    int32_t E, N, D, append_agent_id, ip;  // ip = padded input width
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int H>
__global__ __launch_bounds__(64) void actor_kernel(const ActorArgs a) {
    constexpr int HP = H + 4;  // LDS row pitch (floats): rows 16 B aligned, bank-staggered
    __shared__ __attribute__((aligned(16))) float Xs[TM][MAX_IP + 4];
    __shared__ __attribute__((aligned(16))) float W1s[32][MAX_IP + 4];  // fc1's weights, one 32-column block at a time
    __shared__ __attribute__((aligned(16))) float Y1[TM][HP];
    __shared__ __attribute__((aligned(16))) float Hs[TM][HP];
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
    const int E = a.E, N = a.N, A = a.w.n_actions, I = a.w.input_dim, IP = a.ip;
    const bool shared = a.w.n_sets == 1;
    // tile -> weight set and rows
    int set = 0, base;
    if (shared) {
        base = blockIdx.x * TM;  // rows base .. base+31 of the flat [E*N] row space
    } else {
        const int tiles_per_agent = (E + TM - 1) / TM;
        set = blockIdx.x / tiles_per_agent;
        base = (blockIdx.x - set * tiles_per_agent) * TM;  // envs base .. base+31, agent `set`
    }
    const int R = E * N;
    auto row_of = [&](int i) { return shared ? base + i : (base + i) * N + set; };  // flat row index of tile row i
    auto row_ok = [&](int i) { return shared ? (base + i) < R : (base + i) < E; };
    const float *W1 = a.w.w1 + static_cast<size_t>(set) * H * I, *B1 = a.w.b1 + static_cast<size_t>(set) * H;
    const float *Wih = a.w.wih + static_cast<size_t>(set) * 3 * H * H, *Bih = a.w.bih + static_cast<size_t>(set) * 3 * H;
    const float *Whh = a.w.whh + static_cast<size_t>(set) * 3 * H * H, *Bhh = a.w.bhh + static_cast<size_t>(set) * 3 * H;
    const float *W2 = a.w.w2 + static_cast<size_t>(set) * A * H, *B2 = a.w.b2 + static_cast<size_t>(set) * A;

    // ---- stage the input rows (observation + optional one-hot agent id), the old hidden state and fc1's weights
    for (int idx = lane; idx < TM * IP; idx += 64) {
        const int i = idx / IP, k = idx - i * IP;
        float v = 0.0f;
        if (row_ok(i)) {
            const int r = row_of(i);
            // a restarted env is seen through the reference's reset() observation: zeros (PredatorCapturePrey.py:136)
            if (k < a.D) v = (a.restart && a.restart[r / N] != 0) ? 0.0f : a.obs[static_cast<size_t>(r) * a.D + k];
            else if (a.append_agent_id && k - a.D == r % N) v = 1.0f;
        }
        Xs[i][k] = v;
    }
    for (int idx = lane; idx < TM * (H / 4); idx += 64) {
        const int i = idx / (H / 4), k4 = idx - i * (H / 4);
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (row_ok(i)) {
            const int r = row_of(i);
            const bool fresh = a.restart && a.restart[r / N] != 0;
            if (!fresh) v = *reinterpret_cast<const float4 *>(a.hidden + static_cast<size_t>(r) * H + 4 * k4);
        }
        *reinterpret_cast<float4 *>(&Hs[i][4 * k4]) = v;
    }
    __syncthreads();

    auto zero16 = [] {
        floatx16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.0f;
        return z;
    };
    auto mfma4 = [](floatx16 acc, const float4 &x, const float4 &w) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, w.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, w.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, w.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, w.w, acc, 0, 0, 0);
        return acc;
    };
    auto crow = [&](int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; };  // tile row of accumulator register `reg`

    // ---- fc1 + ReLU: Y1 = relu(X W1^T + b1)
    for (int cb = 0; cb < H / 32; ++cb) {
        const int n = cb * 32 + col;
        if (cb) __syncthreads();  // the previous block's weights have been consumed
        for (int idx = lane; idx < 32 * IP; idx += 64) {  // ragged rows (I is not a multiple of 4): staged zero-padded
            const int nn = idx / IP, k = idx - nn * IP;
            W1s[nn][k] = k < I ? W1[(cb * 32 + nn) * I + k] : 0.0f;
        }
        __syncthreads();
        floatx16 acc = zero16();
        for (int kk = 0; kk < IP / 2; kk += 4) {
            const float4 x = *reinterpret_cast<const float4 *>(&Xs[col][half * (IP / 2) + kk]);
            const float4 w = *reinterpret_cast<const float4 *>(&W1s[col][half * (IP / 2) + kk]);
            acc = mfma4(acc, x, w);
        }
        const float b = B1[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) Y1[crow(r)][n] = fmaxf(acc[r] + b, 0.0f);
    }
    __syncthreads();

    // ---- recurrent layer
    float hn[H / 32][16];  // the new hidden state in accumulator layout (the old one is still an operand)
    if (a.w.use_rnn) {     // torch.nn.GRUCell: gates r, z, n in that order
#pragma unroll
        for (int cb = 0; cb < H / 32; ++cb) {
            const int j = cb * 32 + col;
            floatx16 gi[3], gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                gi[g] = zero16();
                gh[g] = zero16();
            }
            // software-pipelined: the operands of k-step block kk+4 are in flight while block kk multiplies
            // (one wave per SIMD here: nothing else hides the L2 latency of the weight stream)
            struct Ops {
                float4 x, h, wi[3], wh[3];
            };
            auto fetch = [&](int kk) {
                Ops o;
                const int k0 = half * (H / 2) + kk;
                o.x = *reinterpret_cast<const float4 *>(&Y1[col][k0]);
                o.h = *reinterpret_cast<const float4 *>(&Hs[col][k0]);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const size_t wrow = static_cast<size_t>(g * H + j) * H + k0;
                    o.wi[g] = *reinterpret_cast<const float4 *>(Wih + wrow);
                    o.wh[g] = *reinterpret_cast<const float4 *>(Whh + wrow);
                }
                return o;
            };
            Ops cur = fetch(0);
#pragma unroll 2
            for (int kk = 0; kk < H / 2; kk += 4) {
                Ops nxt = cur;
                if (kk + 4 < H / 2) nxt = fetch(kk + 4);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    gi[g] = mfma4(gi[g], cur.x, cur.wi[g]);
                    gh[g] = mfma4(gh[g], cur.h, cur.wh[g]);
                }
                cur = nxt;
            }
            const float bir = Bih[j], biz = Bih[H + j], bin = Bih[2 * H + j];
            const float bhr = Bhh[j], bhz = Bhh[H + j], bhn = Bhh[2 * H + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float rg_ = sigmoidf_((gi[0][r] + bir) + (gh[0][r] + bhr));
                const float zg = sigmoidf_((gi[1][r] + biz) + (gh[1][r] + bhz));
                const float ng = tanhf((gi[2][r] + bin) + rg_ * (gh[2][r] + bhn));
                hn[cb][r] = (1.0f - zg) * ng + zg * Hs[crow(r)][j];
            }
        }
    } else {  // use_rnn = False: h = relu(Linear(x))  (rnn_agent.py:13,27); the weights sit in the wih / bih slots
#pragma unroll
        for (int cb = 0; cb < H / 32; ++cb) {
            const int j = cb * 32 + col;
            floatx16 acc = zero16();
            for (int kk = 0; kk < H / 2; kk += 4) {
                const int k0 = half * (H / 2) + kk;
                acc = mfma4(acc, *reinterpret_cast<const float4 *>(&Y1[col][k0]),
                            *reinterpret_cast<const float4 *>(Wih + static_cast<size_t>(j) * H + k0));
            }
            const float b = Bih[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) hn[cb][r] = fmaxf(acc[r] + b, 0.0f);
        }
    }
    __syncthreads();  // every read of the old hidden state is done
#pragma unroll
    for (int cb = 0; cb < H / 32; ++cb) {
        const int j = cb * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = crow(r);
            Hs[i][j] = hn[cb][r];
            if (row_ok(i)) a.hidden[static_cast<size_t>(row_of(i)) * H + j] = hn[cb][r];
        }
    }
    __syncthreads();

    // ---- fc2: q = h' W2^T + b2 (A <= 32 columns: one tile), then the greedy action per row
    {
        floatx16 acc = zero16();
        const bool n_ok = col < A;
        for (int kk = 0; kk < H / 2; kk += 4) {
            const int k0 = half * (H / 2) + kk;
            float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (n_ok) w = *reinterpret_cast<const float4 *>(W2 + static_cast<size_t>(col) * H + k0);
            acc = mfma4(acc, *reinterpret_cast<const float4 *>(&Hs[col][k0]), w);
        }
        const float b = n_ok ? B2[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) Y1[crow(r)][col] = acc[r] + b;  // Y1 is free again: the q tile, row-major
    }
    __syncthreads();
    if (lane < TM && row_ok(lane)) {
        const int r = row_of(lane);
        float best = Y1[lane][0];
        int arg = 0;
        for (int n = 0; n < A; ++n) {
            const float v = Y1[lane][n];
            if (a.q) a.q[static_cast<size_t>(r) * A + n] = v;
            if (v > best) {  // first maximum, like torch.argmax
                best = v;
                arg = n;
            }
        }
        if (a.actions) a.actions[r] = arg;
    }
}

}  // namespace rg

static thread_local char g_actor_err[256] = "";

extern "C" const char *rg_actor_last_error(void) { return g_actor_err; }

extern "C" int rg_actor_forward(const rg_actor_weights *w, int32_t num_envs, int32_t n_agents, const float *obs,
                                int32_t obs_dim, int32_t append_agent_id, const uint8_t *restart, float *hidden,
                                float *q, int32_t *actions, void *hip_stream) {
    auto fail = [](int code, const char *msg) {
        snprintf(g_actor_err, sizeof(g_actor_err), "%s", msg);
        return code;
    };
    if (!w || !obs || !hidden) return fail(-1, "weights, obs or hidden is NULL");
    if (!w->w1 || !w->b1 || !w->wih || !w->bih || !w->w2 || !w->b2) return fail(-2, "a weight array is NULL");
    if (w->use_rnn && (!w->whh || !w->bhh)) return fail(-2, "GRU weights whh / bhh are NULL");
    if (w->hidden_dim != 64 && w->hidden_dim != 128) return fail(-3, "hidden_dim must be 64 or 128 (the reference's actors)");
    if (w->n_actions < 1 || w->n_actions > 32) return fail(-4, "n_actions must be in 1..32");
    if (w->n_sets != 1 && w->n_sets != n_agents) return fail(-5, "n_sets must be 1 (shared) or n_agents");
    if (num_envs < 1 || n_agents < 1 || obs_dim < 1) return fail(-6, "num_envs, n_agents, obs_dim must be >= 1");
    const int in_dim = obs_dim + (append_agent_id ? n_agents : 0);
    if (in_dim != w->input_dim) return fail(-7, "obs_dim (+ n_agents with append_agent_id) != the actor's input_dim");
    const int ip = (in_dim + 7) / 8 * 8;
    if (ip > rg::MAX_IP) return fail(-8, "input_dim above 64 is not supported");
    if ((reinterpret_cast<uintptr_t>(hidden) | reinterpret_cast<uintptr_t>(w->wih) | reinterpret_cast<uintptr_t>(w->whh) |
         reinterpret_cast<uintptr_t>(w->w2)) & 15u)
        return fail(-9, "hidden, wih, whh and w2 must be 16-byte aligned");
    rg::ActorArgs a;
    a.w = *w;
    a.obs = obs;
    a.restart = restart;
    a.hidden = hidden;
    a.q = q;
    a.actions = actions;
    a.E = num_envs;
    a.N = n_agents;
    a.D = obs_dim;
    a.append_agent_id = append_agent_id;
    a.ip = ip;
    const int tiles = w->n_sets == 1 ? (num_envs * n_agents + rg::TM - 1) / rg::TM
                                     : n_agents * ((num_envs + rg::TM - 1) / rg::TM);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (w->hidden_dim == 64) hipLaunchKernelGGL((rg::actor_kernel<64>), dim3(tiles), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((rg::actor_kernel<128>), dim3(tiles), dim3(64), 0, stream, a);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(-30, hipGetErrorString(err));
    return 0;
}
