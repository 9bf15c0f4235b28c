// actor_mfma.hip -- the EPyMARL recurrent actor (utilities/rnn_agent.py:5-29: fc1 -> ReLU -> GRUCell ->
// fc2, or rnn_ns_agent.py:5-36 with one such network per agent) for all E x N agents of a batch in
// ONE launch: policy inference for on-device evaluation rollouts.
//
// This is the one dense contraction on the path, so it runs on the matrix cores: f32-input MFMA
// (v_mfma_f32_32x32x2_f32, exact f32 products and sums -- the parity bar against the reference's
// float32 modules is 1e-5, which rules the 16x faster bf16 forms out).  One workgroup of H/32
// wavefronts owns a tile of 32 agent rows and carries it through the whole network, wavefront w
// computing the 32 hidden columns [32 w, 32 w + 32) of every layer:
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

// gate nonlinearities on the hardware exponential (v_exp_f32, ~1 ulp on 2^t): absolute error ~1e-7 on
// outputs in [0, 1] / [-1, 1], far inside the 1e-5 parity bar, at a tenth of libm's instruction count
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x)); }

#ifdef RG_ACTOR_STAMPS  // diagnostic build (tools/actor_stamps.py): wave-cycle stamps of the phases, written over q
#define RG_ASTAMP(i) stamps[i] = static_cast<int>(__builtin_amdgcn_s_memtime() - t_start)
#else
#define RG_ASTAMP(i)
#endif

template <int H>
__global__ __launch_bounds__(64 * (H / 32)) void actor_kernel(const ActorArgs a) {
#ifdef RG_ACTOR_STAMPS
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    int stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    constexpr int HP = H + 4;  // LDS row pitch (floats): rows 16 B aligned, bank-staggered
    constexpr int NTHREADS = 64 * (H / 32);
    __shared__ __attribute__((aligned(16))) float Xs[TM][MAX_IP + 4];
    __shared__ __attribute__((aligned(16))) float Y1[TM][HP];
    __shared__ __attribute__((aligned(16))) float Hs[TM][HP];
    const int tid = threadIdx.x, lane = tid & 63, cb = tid >> 6, half = lane >> 5, col = lane & 31;
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
    for (int idx = tid; idx < TM * IP; idx += NTHREADS) {
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
    for (int idx = tid; idx < TM * (H / 4); idx += NTHREADS) {
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
    RG_ASTAMP(0);  // inputs staged

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

    // ---- fc1 + ReLU: Y1 = relu(X W1^T + b1); this wave's 32 columns.  The layer is small and its rows are
    // ragged (I is not a multiple of 4): scalar operand loads, zero beyond I.
    {
        const int n = cb * 32 + col;
        floatx16 acc = zero16();
        float w[MAX_IP / 2], xv[MAX_IP / 2];  // all operand loads in flight before the first product
#pragma unroll
        for (int kk = 0; kk < MAX_IP / 2; ++kk) {
            const int k = half * (IP / 2) + kk;
            const bool in = kk < IP / 2;
            w[kk] = (in && k < I) ? W1[n * I + k] : 0.0f;
            xv[kk] = in ? Xs[col][k] : 0.0f;
        }
#pragma unroll
        for (int kk = 0; kk < MAX_IP / 2; ++kk)
            if (kk < IP / 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[kk], w[kk], acc, 0, 0, 0);
        const float b = B1[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) Y1[crow(r)][n] = fmaxf(acc[r] + b, 0.0f);
    }
    __syncthreads();
    RG_ASTAMP(1);  // fc1

    // ---- recurrent layer
    float hn[16];       // the new hidden state in accumulator layout (the old one is still an operand)
    if (a.w.use_rnn) {  // torch.nn.GRUCell: gates r, z, n in that order
        {
            const int j = cb * 32 + col;
            floatx16 gi[3], gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                gi[g] = zero16();
                gh[g] = zero16();
            }
            // The weight stream: lane (col, half) owns row g H + j of each gate matrix and, of that row, the
            // k range of its half -- consumed in chunks of 32 floats = one 128-byte line per lane, eight
            // float4 loads issued together, so a line is used up while it is hot (with 16 B per visit the
            // 64 lines x 6 matrices a wave walks at once thrash L1 and every visit refetches its line from
            // L2).  Groups (chunk, gate, matrix) are software-pipelined: the next group's line is in flight
            // while this group's 32 MFMAs run.
            constexpr int NCH = (H / 2) / 32;  // chunks per lane half
            constexpr int NGROUPS = NCH * 6;
            const bool packed = a.w.gru_packed != 0;
            auto load_line = [&](int t, float4 (&wl)[8]) {
                const int chunk = t / 6, g = (t % 6) >> 1, hh = t & 1;
                const float *M = hh ? Whh : Wih;
                if (packed) {  // rg_actor_pack_gru order: [cb][chunk][gate][q4][lane][4] -- one contiguous KB per load
                    const float *src = M + (static_cast<size_t>((cb * NCH + chunk) * 3 + g) * 8) * 256 + lane * 4;
#pragma unroll
                    for (int q4 = 0; q4 < 8; ++q4) wl[q4] = *reinterpret_cast<const float4 *>(src + q4 * 256);
                } else {
                    const float *src = M + static_cast<size_t>(g * H + j) * H + half * (H / 2) + chunk * 32;
#pragma unroll
                    for (int q4 = 0; q4 < 8; ++q4) wl[q4] = *reinterpret_cast<const float4 *>(src + 4 * q4);
                }
            };
            float4 wcur[8], wnext[8], xa[8], ha[8];
            load_line(0, wcur);
#pragma unroll
            for (int t = 0; t < NGROUPS; ++t) {
                const int chunk = t / 6, g = (t % 6) >> 1, hh = t & 1;
                if (t % 6 == 0) {  // this chunk's activations: A operands for all six products
                    const int k0 = half * (H / 2) + chunk * 32;
#pragma unroll
                    for (int q4 = 0; q4 < 8; ++q4) {
                        xa[q4] = *reinterpret_cast<const float4 *>(&Y1[col][k0 + 4 * q4]);
                        ha[q4] = *reinterpret_cast<const float4 *>(&Hs[col][k0 + 4 * q4]);
                    }
                }
                if (t + 1 < NGROUPS) load_line(t + 1, wnext);
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) {
                    if (hh) gh[g] = mfma4(gh[g], ha[q4], wcur[q4]);
                    else gi[g] = mfma4(gi[g], xa[q4], wcur[q4]);
                }
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) wcur[q4] = wnext[q4];
            }
#ifdef RG_ACTOR_STAMPS
#pragma unroll
            for (int g = 0; g < 3; ++g) asm volatile("" ::"v"(gi[g]), "v"(gh[g]));
            RG_ASTAMP(2);  // GRU products
#endif
            const float bir = Bih[j], biz = Bih[H + j], bin = Bih[2 * H + j];
            const float bhr = Bhh[j], bhz = Bhh[H + j], bhn = Bhh[2 * H + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float rg_ = sigmoidf_((gi[0][r] + bir) + (gh[0][r] + bhr));
                const float zg = sigmoidf_((gi[1][r] + biz) + (gh[1][r] + bhz));
                const float ng = tanhf_((gi[2][r] + bin) + rg_ * (gh[2][r] + bhn));
                hn[r] = (1.0f - zg) * ng + zg * Hs[crow(r)][j];
            }
        }
    } else {  // use_rnn = False: h = relu(Linear(x))  (rnn_agent.py:13,27); the weights sit in the wih / bih slots
        {
            const int j = cb * 32 + col;
            floatx16 acc = zero16();
            for (int kk = 0; kk < H / 2; kk += 4) {
                const int k0 = half * (H / 2) + kk;
                acc = mfma4(acc, *reinterpret_cast<const float4 *>(&Y1[col][k0]),
                            *reinterpret_cast<const float4 *>(Wih + static_cast<size_t>(j) * H + k0));
            }
            const float b = Bih[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) hn[r] = fmaxf(acc[r] + b, 0.0f);
        }
    }
#ifdef RG_ACTOR_STAMPS
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(hn[r]));
    RG_ASTAMP(3);  // gates
#endif
    __syncthreads();  // every read of the old hidden state is done
    {
        const int j = cb * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = crow(r);
            Hs[i][j] = hn[r];
            if (row_ok(i)) a.hidden[static_cast<size_t>(row_of(i)) * H + j] = hn[r];
        }
    }
    __syncthreads();
    RG_ASTAMP(4);  // new hidden state stored

    // ---- fc2: q = h' W2^T + b2 (A <= 32 columns: one tile, wavefront 0), then the greedy action per row
    if (cb == 0) {
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
    RG_ASTAMP(5);  // fc2
    if (tid < TM && row_ok(tid)) {
        const int r = row_of(tid);
        float best = Y1[tid][0];
        int arg = 0;
        for (int n = 0; n < A; ++n) {
            const float v = Y1[tid][n];
            if (a.q) a.q[static_cast<size_t>(r) * A + n] = v;
            if (v > best) {  // first maximum, like torch.argmax
                best = v;
                arg = n;
            }
        }
        if (a.actions) a.actions[r] = arg;
    }
#ifdef RG_ACTOR_STAMPS
    RG_ASTAMP(6);
    stamps[7] = static_cast<int>(t_start & 0x7FFFFFFF);   // absolute start (low bits): which waves ran side by side
    __syncthreads();
    if (lane == 0 && a.q) {
        int *dst = reinterpret_cast<int *>(a.q) + static_cast<size_t>(E) * N * A + (static_cast<size_t>(blockIdx.x) * (H / 32) + cb) * 8;  // behind the q block
        for (int i = 0; i < 8; ++i) dst[i] = stamps[i];
    }
#endif
}

// torch layout [S][3H][H] -> the kernel's streaming order [S][cb][chunk][gate][q4][lane = (half, col)][4]
__global__ void pack_gru_kernel(const float *src, float *dst, int n_sets, int H) {
    const int nch = (H / 2) / 32, ncb = H / 32;
    const size_t total = static_cast<size_t>(n_sets) * 3 * H * H;
    for (size_t o = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<size_t>(gridDim.x) * blockDim.x) {
        size_t r = o;
        const int f = r % 4; r /= 4;
        const int lane = r % 64; r /= 64;
        const int q4 = r % 8; r /= 8;
        const int g = r % 3; r /= 3;
        const int chunk = r % nch; r /= nch;
        const int cb = r % ncb; r /= ncb;
        const int s = static_cast<int>(r);
        const int half = lane >> 5, col = lane & 31;
        const int row = g * H + cb * 32 + col, k = half * (H / 2) + chunk * 32 + q4 * 4 + f;
        dst[o] = src[(static_cast<size_t>(s) * 3 * H + row) * H + k];
    }
}

}  // namespace rg

static thread_local char g_actor_err[256] = "";

extern "C" int rg_actor_pack_gru(const float *src, int32_t n_sets, int32_t hidden_dim, float *dst, void *hip_stream) {
    if (!src || !dst || n_sets < 1 || (hidden_dim != 64 && hidden_dim != 128)) {
        snprintf(g_actor_err, sizeof(g_actor_err), "rg_actor_pack_gru: NULL array, n_sets < 1 or hidden_dim not 64 / 128");
        return -1;
    }
    hipLaunchKernelGGL(rg::pack_gru_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(hip_stream), src, dst, n_sets,
                       hidden_dim);
    return hipGetLastError() == hipSuccess ? 0 : -30;
}

extern "C" const char *rg_actor_last_error(void) { return g_actor_err; }

#ifdef RG_ACTOR_STAMPS  // diagnostic build: what the runtime says about co-resident workgroups per CU
extern "C" int rg_actor_occupancy(int hidden_dim) {
    int n = -1;
    if (hidden_dim == 64) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rg::actor_kernel<64>, 128, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rg::actor_kernel<128>, 256, 0);
    return n;
}
#endif

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
    if (w->hidden_dim == 64) hipLaunchKernelGGL((rg::actor_kernel<64>), dim3(tiles), dim3(128), 0, stream, a);
    else hipLaunchKernelGGL((rg::actor_kernel<128>), dim3(tiles), dim3(256), 0, stream, a);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(-30, hipGetErrorString(err));
    return 0;
}
