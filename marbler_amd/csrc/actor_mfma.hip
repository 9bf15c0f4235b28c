// actor_mfma.hip -- the EPyMARL recurrent actor (utilities/rnn_agent.py:5-29: fc1 -> ReLU -> GRUCell ->
// fc2, or rnn_ns_agent.py:5-36 with one such network per agent) for all E x N agents of a batch in
// ONE launch: policy inference for on-device evaluation rollouts.
//
// This is the one dense contraction on the path, so it runs on the matrix cores.  One workgroup of H/32 wavefronts owns a
// tile of 32 agent rows and carries it through the whole network, wavefront w computing the 32 hidden columns
// [32 w, 32 w + 32) of every layer:
//     X [32 x I]  --fc1-->  Y [32 x H]  --GRU (6 gate tiles per 32 hidden columns)-->  h' [32 x H]  --fc2--> q [32 x A]
// * The GRU -- 96 % of the arithmetic -- has three forms.  gru_packed == 3 (rg_actor_pack_gru_f16x2, the default of
//   marbler_amd/evaluate.py since round 5): every float32 value is carried as TWO binary16 planes (hi, 2^11 lo) and a float32
//   product becomes three plane products on v_mfma_f32_32x32x16_f16, the cross terms in a second accumulator; the activations are
//   split where they are produced, into plane images in LDS (see "two binary16 planes" below).  gru_packed == 2
//   (rg_actor_pack_gru_bf16x3, round 4): THREE bfloat16 planes (split8 below), six plane products on v_mfma_f32_32x32x16_bf16
//   (32 cycles for K = 16), float32's exponent range, an error below a float32 dot product's own roundings.  gru_packed == 0 / 1:
//   f32-input MFMA (v_mfma_f32_32x32x2_f32, 64 cycles for K = 2), exact float32 products.  At 4096 x 4 rows, hidden 128:
//   19-20 / 24-26 / 44-50 us per launch.
// * A operands (activations) are read from LDS; B operands of the GRU (weights) stream from L2 in the order a pack routine
//   wrote them (1 KB per load instruction), a ring of groups ahead of the MFMAs, held in place by scheduling fences.  fc1's
//   small ragged operands are fetched once per tile, coalesced, and staged in LDS (inputs up to 32 wide; wider ones are read
//   as they lie by a ROLLED loop: unrolled, the layer was ~600 instructions of cold code at the head of every launch).
// * Two [32][H] float32 LDS images with an XOR swizzle instead of padding (a third, of binary16 planes, in the two-plane form:
//   48 KB per tile at hidden 128), and 256 registers per lane (amdgpu_waves_per_eu): TWO tiles per CU, one wave of each per SIMD.
//   (Rounds 1-3 ran one tile per CU -- 352 registers per lane; LDS was never the limit: hipOccupancyMaxActiveBlocksPerMultiprocessor
//   reckons with 64 KB of LDS per CU, the hardware has 160.)
// * Memory traffic is ordered by hand: the head requests the restart flag, fc1's staged operands and then the hidden state (the
//   vector-memory counter retires in order: the wait in front of the staging stores leaves the hidden state in flight); the GRU's
//   first weight groups go out ahead of fc1's epilogue; the new hidden state is stored right after the gates and drains under fc2
//   and the arg-max, whose barriers order LDS only (lds_barrier).
// * Layer outputs come out of the MFMA with the column on the lane and 16 rows in registers (C/D map: col = lane & 31,
//   row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)); the gate arithmetic is elementwise in that layout and one LDS write turns it
//   into the next layer's A image.  fc2's K range is split over the tile's wavefronts, the partial tiles meet in LDS, and all
//   threads take part in the bias / arg-max / q stores.  Non-shared actors: a tile takes the rows of ONE agent index (stride N),
//   so the whole tile uses one weight set.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/robogym.h"
#include "probes/actor_diag.h"   // RG_ASTAMP* / RG_AKEEP*: phase stamps of the diagnostic builds, nothing in the shipped one

namespace rg {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- float32 carried as three bfloat16 planes (gru_packed == 2; round 4).
// v_mfma_f32_32x32x2_f32 costs 64 cycles for K = 2; v_mfma_f32_32x32x16_bf16 costs 32 cycles for K = 16: 16 x the rate.  A
// float32 value is EXACTLY hi + mid + lo with hi = the top 16 bits of x (a bfloat16 by truncation), mid = the top 16 bits of
// x - hi, lo = the top 16 bits of x - hi - mid (what is dropped is below 2^-24 |x|).  A product x w is then the sum of nine
// plane products, of which the six of order >= 2^-16 are computed -- hh, hm, mh, hl, lh, mm, each exact in the MFMA's float32
// accumulator -- and the three of order 2^-24 are left out: 6 / 16 of the float32 MFMA time for an error BELOW that of a
// float32 dot product's own roundings (measured on 4096 x 128 x 384 random operands: 3.6e-7 against the exact product, a
// float32 GEMM 2.5e-6).  Same exponent range as float32: nothing can overflow or flush that float32 would not.
// split8: eight consecutive k values of one row -> the three planes as MFMA operands (element e of a plane = k0 + e)
__device__ __forceinline__ void split8(const float4 &lo4, const float4 &hi4, bf16x8 &ph, bf16x8 &pm, bf16x8 &pl) {
    const float x[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    uint32_t bh[8], bm[8], bl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bh[e] = __builtin_bit_cast(uint32_t, x[e]) & 0xFFFF0000u;
        const float r1 = x[e] - __builtin_bit_cast(float, bh[e]);   // exact
        bm[e] = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, bm[e]);     // exact
        bl[e] = __builtin_bit_cast(uint32_t, r2);                    // (truncated by the packing below)
    }
    u32x4 h, m, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // two bfloat16 per register: element 2q in the low half
        h[q] = __builtin_amdgcn_perm(bh[2 * q + 1], bh[2 * q], 0x07060302u);
        m[q] = __builtin_amdgcn_perm(bm[2 * q + 1], bm[2 * q], 0x07060302u);
        l[q] = __builtin_amdgcn_perm(bl[2 * q + 1], bl[2 * q], 0x07060302u);
    }
    ph = __builtin_bit_cast(bf16x8, h);
    pm = __builtin_bit_cast(bf16x8, m);
    pl = __builtin_bit_cast(bf16x8, l);
}

// ---- float32 carried as TWO binary16 planes (gru_packed == 3; round 5): half the matrix-core time of the three-plane form.
// x = hi + lo with hi = x rounded to binary16 (11 significant bits) and lo = x - hi (exact, below 2^-11 |x|); lo is
// carried SCALED by 2^11 -- lo' = binary16(2048 lo), again 11 significant bits, well inside binary16's exponent range wherever
// hi is -- so what is dropped is below 2^-22 |x|.  A product x w is hi hi + (hi lo' + lo' hi) 2^-11 + O(2^-22): THREE plane
// products on v_mfma_f32_32x32x16_f16 (the same 32 cycles for K = 16 as the bfloat16 instruction), the two cross products into
// a second accumulator that joins the first with one multiply-add in the gate arithmetic.  Measured on 2048 x 128 x 384 random
// operands against the exact product: max 3.9e-7, rms 5.6e-8 -- a float32 GEMM of the same operands: 9.5e-7 / 7.5e-8.
// Range: binary16's.  Activations are the hidden state (in [-1, 1]) and fc1's ReLU output; |x| > 65504 saturates (f16_saturate:
// never inf; the same for a weight).
// Below 2^-14 hi is a binary16 DENORMAL (spacing 2^-24) and lo' the 11 bits after it: conversions and the matrix cores take
// denormal operands as they are on gfx950 (round 5: with them flushed -- s_setreg MODE.FP_DENORM -- a hidden state of 3e-5 kept 11
// bits in all, and products against large weights were off by 2e-4; tests/test_gpu_actor.py holds the case).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float F16_LO_SCALE = 2048.0f, F16_LO_UNSCALE = 1.0f / 2048.0f;
// The activations are split where they are PRODUCED (fc1's epilogue, the hidden state's staging copy: split1 below), once per tile,
// into plane images in LDS; the products' loop reads 16-byte operands and runs no conversion (split by every wave at every K step,
// 64 conversion instructions per step stood between the MFMAs: 52 cycles per MFMA against the pipe's 32).

constexpr int TM = 32;        // agent rows per wavefront
constexpr int MAX_IP = 64;    // padded input width (multiple of 8)

struct ActorArgs {
    rg_actor_weights w;
    const float *obs;       // [E][N][D]
    const uint8_t *restart; // [E] or NULL: nonzero = a new episode: hidden state and observation are taken as zero
    float *hidden;          // [E][N][H] in/out
    float *q;               // [E][N][A] or NULL
    int32_t *actions;       // [E][N] or NULL
    const float *explore_u; // [E][N] uniforms in [0, 1) or NULL: epsilon-greedy selection (rg_actor_forward_explore)
    float explore_scale;    // n_actions / epsilon
    int32_t E, N, D, append_agent_id, ip;  // ip = padded input width
};

// gate nonlinearities on the hardware exponential (v_exp_f32, ~1 ulp on 2^t): absolute error ~1e-7 on
// outputs in [0, 1] / [-1, 1], far inside the 1e-5 parity bar, at a tenth of libm's instruction count
// (v_rcp_f32 is within 1 ulp; `1.0f / x` would be the ten-instruction correctly rounded division, 48 times per lane and tile)
// torch.relu: a NaN stays a NaN (v_max_f32 would return the 0)
__device__ __forceinline__ float relu_(float x) { return x < 0.0f ? 0.0f : x; }
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }


// Two [TM][H] images in LDS with pitch exactly H.  Bank conflicts are avoided by an XOR swizzle of the 16-byte block index
// with the row instead of padding: block b of row i lives at block b ^ (i & 7).  The eight lanes of an LDS lane group read the
// same logical block of eight consecutive rows -> eight different physical blocks -> all 32 banks.
template <int H>
__device__ __forceinline__ int swz(int i, int k) { return i * H + ((((k >> 2) ^ (i & 7)) << 2) | (k & 3)); }
template <int H>
__device__ __forceinline__ int swz4(int i, int k4) { return i * H + ((k4 ^ (i & 7)) << 2); }

// binary16 plane images [TM][H] (pitch H halves): the 16-byte block b8 = k / 8 of row i lives at block b8 ^ (i & 7) -- the eight lanes
// of an LDS lane group read one logical block of eight consecutive rows: eight physical blocks, all banks
template <int H>
__device__ __forceinline__ int swz8(int i, int b8) { return i * H + ((b8 ^ (i & 7)) << 3); }
// one value -> its two planes, rounded to nearest (the activations are split where they are PRODUCED, once per tile)
// (beyond binary16's largest finite value the conversion would round to infinity and the low plane to NaN: the value saturates
// instead -- comparisons, so that a NaN stays a NaN)
constexpr float F16_MAX = 65504.0f;
__device__ __forceinline__ float f16_saturate(float x) { return x > F16_MAX ? F16_MAX : x < -F16_MAX ? -F16_MAX : x; }
__device__ __forceinline__ void split1(float x, _Float16 &hi, _Float16 &lo) {
    x = f16_saturate(x);
    hi = static_cast<_Float16>(x);
    lo = static_cast<_Float16>((x - static_cast<float>(hi)) * F16_LO_SCALE);
}

// a workgroup barrier that orders LDS traffic only: global stores in flight stay in flight
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int H, int SPLIT>   // SPLIT: the GRU's products on 1: three bfloat16 planes (gru_packed == 2), 2: two binary16 planes (gru_packed == 3)
__attribute__((amdgpu_waves_per_eu(2, 2)))   // 256 registers (VGPR + AGPR): two tiles per CU, one's serial phases under the other's MFMAs
__global__ __launch_bounds__(64 * (H / 32)) void actor_kernel(const ActorArgs a) {
    RG_ASTAMP_BEGIN();
    constexpr int NW = H / 32;  // wavefronts per tile
    constexpr int NTHREADS = 64 * NW;
    // SPLIT == 2: a third image.  Y is then held as its two binary16 planes (in the place of its float32 image: the same size), the
    // old hidden state as float32 (the blend with the new one wants it exact) AND as planes in the third image
    __shared__ __attribute__((aligned(16))) float lds[(SPLIT == 2 ? 3 : 2) * TM * H];
    float *const Y = lds;            // fc1's output (A operand of the GRU), later fc2's partial sums
    float *const Hs = lds + TM * H;  // the old hidden state (A operand), then the new one
    _Float16 *const Yp = reinterpret_cast<_Float16 *>(lds);                // planes: hi at 0, lo' at TM * H halves
    _Float16 *const Hp = reinterpret_cast<_Float16 *>(lds + (SPLIT == 2 ? 2 : 0) * TM * H);
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
    // (use_rnn = 0: the wih / bih slots hold ONE H x H layer per set, not three gates -- a non-shared MLP actor, the reference's
    // mappo_ns, read past its arrays with the GRU's stride until round 4's shape fuzz)
    const int GR = a.w.use_rnn ? 3 * H : H;
    const float *Wih = a.w.wih + static_cast<size_t>(set) * GR * H, *Bih = a.w.bih + static_cast<size_t>(set) * GR;
    const float *Whh = a.w.whh + static_cast<size_t>(set) * 3 * H * H, *Bhh = a.w.bhh + static_cast<size_t>(set) * 3 * H;
    const float *W2 = a.w.w2 + static_cast<size_t>(set) * A * H, *B2 = a.w.b2 + static_cast<size_t>(set) * A;

    // ---- the old hidden state: every thread's share of the tile is requested at the head and lands in LDS behind fc1.
    // ORDER of the head's requests (the vector-memory counter retires in order, a wait can only leave the YOUNGEST loads in
    // flight): restart flag, fc1's staged operands, THEN the hidden state -- so that the wait in front of the staging stores
    // leaves the hidden state's trip to HBM in flight.  (Until round 5 the hidden state was asked for first, and the restart
    // flag's `s_waitcnt vmcnt(0)` -- the flag goes into LDS ahead of the staging loads -- waited for all of it: the staging loads
    // were not even issued before the slowest load of the launch had come back.)
    constexpr int HV = (TM * (H / 4)) / NTHREADS;  // float4 per thread (4)
    float4 hv[HV];
    auto request_hidden = [&] {
#pragma unroll
        for (int m = 0; m < HV; ++m) {
            const int idx = tid + NTHREADS * m, i = idx / (H / 4), k4 = idx % (H / 4);
            // (the loads do not wait for the restart flag: one memory round trip, the flag is applied to what comes back)
            const int r = row_ok(i) ? row_of(i) : 0;
            hv[m] = *reinterpret_cast<const float4 *>(a.hidden + static_cast<size_t>(r) * H + 4 * k4);
        }
    };
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

    // ---- fc1 + ReLU: Y = relu(X W1^T + b1); this wave's 32 columns.
    // The MFMA wants its operands one ROW per lane (A: observation row `col`, B: weight row n), and read that way from memory
    // every load instruction touches 32 rows = 32 cache lines for 4 bytes each, by every wave of the tile again: ~6 000 line
    // look-ups per tile, and the wave spent 8 k of its 45 k cycles in this small layer waiting on the CU's L1 (head stamps,
    // round 4).  So for inputs up to 32 wide the tile's X block and each wave's 32 rows of W1 are fetched ONCE, coalesced
    // (TPRX neighbouring lanes share a row; W1 as a flat run of float4), and the operands are read from LDS.  X is staged as
    // the layer's final input -- restart flag applied, one-hot agent id appended, zero beyond I -- in what becomes Y; the W1
    // chunks in what becomes the old hidden state's image.  Wider inputs (up to MAX_IP) read memory directly, as before.
    constexpr int TPRX = NTHREADS / TM;   // threads per tile row in the staging pass (8 or 4)
    constexpr int XP = 33;                // staged X pitch in floats (odd: the 32 rows fall into 32 banks)
    constexpr int WCH = TM * 32;          // floats per wave's W1 chunk (32 rows x at most 32 inputs)
    static_assert(TM * XP + TM <= TM * H && (H / 32) * WCH <= TM * H, "the staging areas must fit the two LDS images");
    float *const Xs = Y;
    int *const live_s = reinterpret_cast<int *>(Y + TM * XP);   // per tile row: 1 = takes its observation and hidden state
    float *const Ws = Hs + cb * WCH;
    const bool staged = IP <= 32;
    const int n = cb * 32 + col;
    floatx16 acc = zero16();
    const float b1 = B1[n];
    {
        // this thread's tile row in the staging pass
        const int srow = tid / TPRX, part = tid % TPRX;
        const bool ok = row_ok(srow);
        const int r = ok ? row_of(srow) : 0;
        const int env = shared ? r / N : base + srow, agent = shared ? r - env * N : set;
        int restarted = a.restart ? a.restart[ok ? env : 0] : 0;
        const float *xrow = a.obs + static_cast<size_t>(r) * a.D;
        const int id_k = (ok && a.append_agent_id) ? a.D + agent : -1;   // where this row's one-hot agent id sits
        float xv[32 / TPRX];
        u32x4 wv[4];
        const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(W1 + static_cast<size_t>(cb) * 32 * I);   // rows 32 cb .. 32 cb + 31: 8 I float4
        if (staged) {
            // (unconditional loads from clamped places, what lies beyond the row is replaced where it is used: a predicated load is
            // a branch around one instruction, twelve of them here)
#pragma unroll
            for (int m = 0; m < 32 / TPRX; ++m) {
                const int k = part + TPRX * m;
                xv[m] = xrow[k < a.D ? k : 0];   // (not gated by the restart flag: that would be a second round trip)
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int f4 = lane + 64 * m;
                wv[m] = wsrc[f4 < 8 * I ? f4 : 0];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        RG_HSTAMP(3);  // (head) staging loads requested
        request_hidden();
        __builtin_amdgcn_sched_barrier(0);
        RG_HSTAMP(2);  // (head) hidden-state loads issued
        // a restarted env starts from the reference's reset(): zero hidden state, zero observation (PredatorCapturePrey.py:136)
        // (the flag is kept opaque up to here: left to itself the compiler compares it to zero where it is loaded, and the wait for
        // the load stands in front of every other request of the head)
        asm volatile("" : "+v"(restarted));
        const bool live = ok && restarted == 0;
        if (part == 0) live_s[srow] = live ? 1 : 0;
        if (staged) {
#pragma unroll
            for (int m = 0; m < 32 / TPRX; ++m) {
                const int k = part + TPRX * m;
                Xs[srow * XP + k] = k < a.D ? (live ? xv[m] : 0.0f) : (k == id_k ? 1.0f : 0.0f);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int f4 = lane + 64 * m;
                if (f4 < 8 * I) *reinterpret_cast<u32x4 *>(Ws + 4 * f4) = wv[m];
            }
        }
    }
    __syncthreads();   // X, the flags (and, wave by wave, the W1 chunks) are in LDS
    int keep[HV];      // (read now: the flags' place is overwritten by Y below)
#pragma unroll
    for (int m = 0; m < HV; ++m) keep[m] = live_s[(tid + NTHREADS * m) / (H / 4)];
    const int steps = IP / 2;   // per lane half; IP is a multiple of 8
    if (staged) {
#pragma unroll 1
        for (int kk0 = 0; kk0 < steps; kk0 += 4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = half * steps + kk0 + c;
                const float x = Xs[col * XP + k];
                const float w = k < I ? Ws[col * I + k] : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, w, acc, 0, 0, 0);
            }
        }
    } else {
        // scalar operand loads straight from memory (the observation row of tile row `col` + the one-hot agent id), zero beyond I
        const bool ok = row_ok(col);
        const int r = ok ? row_of(col) : 0;
        const int env = r / N, agent = r - env * N;
        const float *xrow = a.obs + static_cast<size_t>(r) * a.D;
        const float *wrow = W1 + static_cast<size_t>(n) * I;
        const bool live = live_s[col] != 0;
        const int id_k = (ok && a.append_agent_id) ? a.D + agent : -1;
        // A ROLLED loop, four k-steps per trip: unrolled over the widest input (32 steps, two predicated loads each) the layer was
        // ~600 instructions of straight-line code that every CU fetches cold at the start of every launch (round 4)
#pragma unroll 1
        for (int kk0 = 0; kk0 < steps; kk0 += 4) {
            float xq[4], wq[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = half * steps + kk0 + c;
                xq[c] = k < a.D ? xrow[k] : 0.0f;        // (k < D: inside the row; the select below applies the restart flag)
                wq[c] = k < I ? wrow[k] : 0.0f;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = half * steps + kk0 + c;
                const float x = k < a.D ? (live ? xq[c] : 0.0f) : (k == id_k ? 1.0f : 0.0f);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, wq[c], acc, 0, 0, 0);
            }
        }
    }
    RG_HKEEP2(acc, b1);
    RG_HSTAMP(4);  // (head) fc1's products and its bias have arrived
    __syncthreads();   // every wave is done with the staged operands: their place becomes Y and the old hidden state
    // SPLIT == 2: the head of the GRU's weight stream and its biases are requested HERE, ahead of fc1's epilogue (its conversions,
    // LDS stores and barrier), not behind it: the stream's first trip to L2 runs under the epilogue.
    // The stream: [cb][ks][gate][plane][lane][8] binary16, groups (ks, gate, matrix) of two 16-byte operands and three MFMAs; PD
    // groups in flight ahead, held in place by the fences (see the three-plane form).  (3 and 4 in flight measured no faster.)
    constexpr int KS = H / 16, NG = KS * 6, PD = 2, RING = PD + 1;
    u32x4 wq[RING][2];
    float br = 0.0f, bz = 0.0f, bin = 0.0f, bhn = 0.0f;
    const uint16_t *Pih = reinterpret_cast<const uint16_t *>(a.w.wih) + static_cast<size_t>(set) * 3 * H * H * 2;
    const uint16_t *Phh = reinterpret_cast<const uint16_t *>(a.w.whh) + static_cast<size_t>(set) * 3 * H * H * 2;
    auto load_w = [&](int t, u32x4 (&wl)[2]) {
        const int ks = t / 6, g = (t % 6) >> 1, hh = t & 1;
        const uint16_t *src = (hh ? Phh : Pih) + ((static_cast<size_t>((cb * KS + ks) * 3 + g) * 2) * 64 + lane) * 8;
        wl[0] = *reinterpret_cast<const u32x4 *>(src);
        wl[1] = *reinterpret_cast<const u32x4 *>(src + 64 * 8);
    };
    if constexpr (SPLIT == 2) {
        const int j = cb * 32 + col;
        br = Bih[j] + Bhh[j], bz = Bih[H + j] + Bhh[H + j], bin = Bih[2 * H + j], bhn = Bhh[2 * H + j];
#pragma unroll
        for (int t = 0; t < PD; ++t) load_w(t, wq[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SPLIT == 2) {   // the GRU's A operands as binary16 planes, split here once (not by every wave at every k step)
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_) {
            _Float16 hi, lo;
            split1(relu_(acc[r_] + b1), hi, lo);
            const int at = swz8<H>(crow(r_), n >> 3) + (n & 7);
            Yp[at] = hi;
            Yp[TM * H + at] = lo;
        }
    } else {
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_) Y[swz<H>(crow(r_), n)] = relu_(acc[r_] + b1);
    }
#pragma unroll
    for (int m = 0; m < HV; ++m) {
        const int idx = tid + NTHREADS * m, i = idx / (H / 4), k4 = idx % (H / 4);
        const float4 hval = keep[m] ? hv[m] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *reinterpret_cast<float4 *>(&Hs[swz4<H>(i, k4)]) = hval;
        if constexpr (SPLIT == 2) {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 hi, lo;
            _Float16 a_, b_;
            split1(hval.x, a_, b_), hi[0] = a_, lo[0] = b_;
            split1(hval.y, a_, b_), hi[1] = a_, lo[1] = b_;
            split1(hval.z, a_, b_), hi[2] = a_, lo[2] = b_;
            split1(hval.w, a_, b_), hi[3] = a_, lo[3] = b_;
            const int at = swz8<H>(i, k4 >> 1) + 4 * (k4 & 1);
            *reinterpret_cast<f16x4 *>(&Hp[at]) = hi;
            *reinterpret_cast<f16x4 *>(&Hp[TM * H + at]) = lo;
        }
    }
    RG_HSTAMP(5);  // (head) Y and the old hidden state written to LDS
    __syncthreads();
    RG_ASTAMP(1);  // fc1 done, old hidden state staged

    // fc2's operands are requested behind the recurrent layer's products, ahead of its elementwise tail: asked for where they
    // are used, behind two barriers, each is a trip to L2 with nothing to hide it
    float4 w2v[4];
    constexpr int TPR = NTHREADS / TM, CPT = 32 / TPR;   // arg-max pass: TPR threads per tile row, CPT action columns each
    float b2v[CPT];
    auto request_fc2 = [&] {
        // (unconditional loads from clamped rows, the padding columns are zeroed where they are used: a predicated load is a
        // branch, and the compiler put a wait for the loads behind its join)
        const int c2 = col < A ? col : 0;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) w2v[q4] = *reinterpret_cast<const float4 *>(W2 + static_cast<size_t>(c2) * H + cb * 32 + half * 16 + 4 * q4);
#pragma unroll
        for (int c_ = 0; c_ < CPT; ++c_) {
            const int c = (tid % TPR) * CPT + c_;
            b2v[c_] = B2[c < A ? c : 0];
        }
    };

    // ---- recurrent layer
    float hn[16];       // the new hidden state in accumulator layout (the old one is still an operand)
    if (a.w.use_rnn && SPLIT == 2) {   // two binary16 planes: three products per float32 product
        if constexpr (SPLIT == 2) {
            const int j = cb * 32 + col;
            // accumulators: 0 = r, 1 = z (input and recurrent products meet in one), 2 = the input half of n, 3 = its recurrent
            // half (r multiplies that one); am: hi hi, ac: the cross products at scale 2^11
            // (the hi hi accumulators START at the biases: the gate arithmetic -- VALU-bound with both tiles of a SIMD in it -- is four
            // additions per element shorter)
            floatx16 am[4], ac[4];
            const float bias4[4] = {br, bz, bin, bhn};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int r = 0; r < 16; ++r) am[g][r] = bias4[g];
                ac[g] = zero16();
            }
            f16x8 yh, yl, hh_, hl_;
#pragma unroll
            for (int t = 0; t < NG; ++t) {
                const int ks = t / 6, g = (t % 6) >> 1, hh = t & 1;
                if (t + PD < NG) load_w(t + PD, wq[(t + PD) % RING]);
                __builtin_amdgcn_sched_barrier(0);
                if (t % 6 == 0) {   // this step's activations for all six groups: k = 16 ks + 8 half + e of tile row `col`
                    const int at = swz8<H>(col, 2 * ks + half);
                    yh = *reinterpret_cast<const f16x8 *>(&Yp[at]);
                    yl = *reinterpret_cast<const f16x8 *>(&Yp[TM * H + at]);
                    hh_ = *reinterpret_cast<const f16x8 *>(&Hp[at]);
                    hl_ = *reinterpret_cast<const f16x8 *>(&Hp[TM * H + at]);
                }
                const f16x8 wh = __builtin_bit_cast(f16x8, wq[t % RING][0]), wl = __builtin_bit_cast(f16x8, wq[t % RING][1]);
                const f16x8 xh = hh ? hh_ : yh, xl = hh ? hl_ : yl;
                const int ai = g < 2 ? g : 2 + hh;
                ac[ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, ac[ai], 0, 0, 0);
                ac[ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, ac[ai], 0, 0, 0);
                am[ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, am[ai], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) RG_AKEEP2(am[g], ac[g]);
            RG_PSTAMP(2);  // GRU products
            request_fc2();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float rg_ = sigmoidf_(__builtin_fmaf(ac[0][r], F16_LO_UNSCALE, am[0][r]));
                const float zg = sigmoidf_(__builtin_fmaf(ac[1][r], F16_LO_UNSCALE, am[1][r]));
                const float ni = __builtin_fmaf(ac[2][r], F16_LO_UNSCALE, am[2][r]);
                const float nh = __builtin_fmaf(ac[3][r], F16_LO_UNSCALE, am[3][r]);
                // (fused forms written out: this file is compiled with contraction off, and the gates are VALU-bound)
                const float e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(rg_, nh, ni) * 2.8853900817779268f);   // exp(2 a) = 2^(2 a log2 e)
                const float ng = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e2), 1.0f);                // tanh a
                hn[r] = __builtin_fmaf(zg, Hs[swz<H>(crow(r), j)] - ng, ng);                                    // (1 - z) n + z h
            }
        }
    } else if (a.w.use_rnn) {  // torch.nn.GRUCell: gates r, z, n in that order
        {
            const int j = cb * 32 + col;
            floatx16 gi[3], gh[3];
            float bir, biz, bin, bhr, bhz, bhn;
            auto request_biases = [&] {
                bir = Bih[j], biz = Bih[H + j], bin = Bih[2 * H + j];
                bhr = Bhh[j], bhz = Bhh[H + j], bhn = Bhh[2 * H + j];
            };
            if constexpr (SPLIT == 1) {   // requested ahead of the products: behind them the gates would start with a trip to L2
                request_biases();    // (the float32-MFMA form has no registers to spare for that)
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                gi[g] = zero16();
                gh[g] = zero16();
            }
            if constexpr (SPLIT == 1) {
                // K in steps of 16: lane (col, half) holds k = 16 ks + 8 half + e, e = 0..7, of its row (activations: tile row
                // `col`; weights: gate row g H + j) -- three 16-byte operands per matrix and step, written in exactly this
                // order by rg_actor_pack_gru_bf16x3: [cb][ks][gate][plane][lane][8].  Groups (ks, gate, matrix) are
                // software-pipelined two deep: 2 x 6 MFMAs = 384 cycles cover the weight loads' trip to L2.
                constexpr int KS = H / 16, NG = KS * 6;
                // (6 bytes per weight: a set's planes are 3/2 the size of its float32 matrix)
                const size_t set_off = static_cast<size_t>(set) * 3 * H * H * 3;
                const uint16_t *Pih = reinterpret_cast<const uint16_t *>(a.w.wih) + set_off, *Phh = reinterpret_cast<const uint16_t *>(a.w.whh) + set_off;
                auto load_w = [&](int t, u32x4 (&wl)[3]) {
                    const int ks = t / 6, g = (t % 6) >> 1, hh = t & 1;
                    const uint16_t *src = (hh ? Phh : Pih) + ((static_cast<size_t>((cb * KS + ks) * 3 + g) * 3) * 64 + lane) * 8;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wl[pl] = *reinterpret_cast<const u32x4 *>(src + pl * 64 * 8);
                };
                // PD groups in flight ahead of the one being multiplied.  NOTHING may be scheduled across the fences below: left to
                // itself the compiler sinks every weight load to just above its first use (fewer live registers) and the stream
                // waits a trip to L2 per group -- the ring the source described did not exist in the ISA until round 4 looked
                // (s_waitcnt vmcnt(0..3) before every group; a single wave per SIMD took 56 cycles per MFMA, now 45; two waves
                // 40 -> 35 against the pipe's 32).  A mask that lets ALU / MFMA / DS instructions cross lets the loads cross too.
                constexpr int PD = 2, RING = PD + 1;
                constexpr int FENCE = 0;
                u32x4 wq[RING][3];
#pragma unroll
                for (int t = 0; t < PD; ++t) load_w(t, wq[t]);
                __builtin_amdgcn_sched_barrier(FENCE);
                bf16x8 yh, ym, yl, hh_, hm_, hl_;
#pragma unroll
                for (int t = 0; t < NG; ++t) {
                    const int ks = t / 6, g = (t % 6) >> 1, hh = t & 1;
                    if (t + PD < NG) load_w(t + PD, wq[(t + PD) % RING]);
                    __builtin_amdgcn_sched_barrier(FENCE);
                    if (t % 6 == 0) {   // this step's activations, split once for all six products
                        const int k4 = 4 * ks + 2 * half;
                        split8(*reinterpret_cast<const float4 *>(&Y[swz4<H>(col, k4)]), *reinterpret_cast<const float4 *>(&Y[swz4<H>(col, k4 + 1)]), yh, ym, yl);
                        split8(*reinterpret_cast<const float4 *>(&Hs[swz4<H>(col, k4)]), *reinterpret_cast<const float4 *>(&Hs[swz4<H>(col, k4 + 1)]), hh_, hm_, hl_);
                    }
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[t % RING][0]), wm = __builtin_bit_cast(bf16x8, wq[t % RING][1]),
                                 wl = __builtin_bit_cast(bf16x8, wq[t % RING][2]);
                    const bf16x8 xh = hh ? hh_ : yh, xm = hh ? hm_ : ym, xl = hh ? hl_ : yl;
                    floatx16 acc = hh ? gh[g] : gi[g];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, wm, acc, 0, 0, 0);   // small terms first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, acc, 0, 0, 0);
                    if (hh) gh[g] = acc;
                    else gi[g] = acc;
                }
            } else if constexpr (SPLIT == 0) {
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
                    const int k40 = (half * (H / 2) + chunk * 32) >> 2;
#pragma unroll
                    for (int q4 = 0; q4 < 8; ++q4) {
                        xa[q4] = *reinterpret_cast<const float4 *>(&Y[swz4<H>(col, k40 + q4)]);
                        ha[q4] = *reinterpret_cast<const float4 *>(&Hs[swz4<H>(col, k40 + q4)]);
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
            }  // !SPLIT
#pragma unroll
            for (int g = 0; g < 3; ++g) RG_AKEEP2(gi[g], gh[g]);
            RG_PSTAMP(2);  // GRU products
            if constexpr (SPLIT != 1) request_biases();
            request_fc2();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float rg_ = sigmoidf_((gi[0][r] + bir) + (gh[0][r] + bhr));
                const float zg = sigmoidf_((gi[1][r] + biz) + (gh[1][r] + bhz));
                const float ng = tanhf_((gi[2][r] + bin) + rg_ * (gh[2][r] + bhn));
                hn[r] = (1.0f - zg) * ng + zg * Hs[swz<H>(crow(r), j)];
            }
        }
    } else {  // use_rnn = False: h = relu(Linear(x))  (rnn_agent.py:13,27); the weights sit in the wih / bih slots
        {
            const int j = cb * 32 + col;
            floatx16 acc = zero16();
            // the lane's half row of the layer, requested whole (H / 8 float4 = 16 or 8 registers quads) before the first product:
            // asked for step by step, every four MFMAs waited a trip to L2
            float4 wrow[H / 8];
#pragma unroll
            for (int q = 0; q < H / 8; ++q) wrow[q] = *reinterpret_cast<const float4 *>(Wih + static_cast<size_t>(j) * H + half * (H / 2) + 4 * q);
            const float b = Bih[j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < H / 8; ++q)
                acc = mfma4(acc, *reinterpret_cast<const float4 *>(&Y[swz4<H>(col, (half * (H / 2) + 4 * q) >> 2)]), wrow[q]);
            request_fc2();
#pragma unroll
            for (int r = 0; r < 16; ++r) hn[r] = relu_(acc[r] + b);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) RG_AKEEP1(hn[r]);
    RG_PSTAMP(3);  // gates
    // The new hidden state goes out to memory NOW, from the accumulator layout (per register: two rows x 128 contiguous bytes per
    // wave), and drains under fc2 and the arg-max.  The barriers from here on order LDS only (lds_barrier): __syncthreads would
    // wait for these stores at every one of them -- which is why, until round 5, they were issued at the very end, where all 512
    // tiles' 8.4 MB met the memory system in one burst with nothing left to run under it.
    {
        const int j = cb * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = crow(r);
            if (row_ok(i)) a.hidden[static_cast<size_t>(row_of(i)) * H + j] = hn[r];
        }
    }
    lds_barrier();  // every read of the old hidden state and of Y is done
    {
        const int j = cb * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = crow(r);
            Hs[swz<H>(i, j)] = hn[r];
        }
    }
    lds_barrier();
    RG_PSTAMP(4);  // new hidden state in LDS

    // ---- fc2: q = h' W2^T + b2 (A <= 32 columns: one 32 x 32 tile).  The K range is split over the tile's wavefronts --
    // 32 k values = 16 MFMAs each instead of H / 2 on one wavefront while the others wait -- and the partial tiles meet in LDS.
    {
        floatx16 acc = zero16();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            const int k0 = cb * 32 + half * 16 + kk;
            acc = mfma4(acc, *reinterpret_cast<const float4 *>(&Hs[swz4<H>(col, k0 >> 2)]), col < A ? w2v[kk >> 2] : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[cb * (TM * 32) + crow(r) * 32 + col] = acc[r];  // Y is free again: NW partial tiles, row-major
    }
    lds_barrier();
    RG_PSTAMP(5);  // fc2 partial products
    {   // every thread: TPR threads per tile row, CPT action columns each; then the greedy action of the row
        const int i = tid / TPR, sub = tid % TPR;
        const bool ok = row_ok(i);
        const int r = ok ? row_of(i) : 0;
        // torch.argmax's answer on every row, non-finite ones included: the first maximum, a NaN counting as the largest value
        // (the first NaN wins); a row of -inf only keeps the sentinel and is given column 0 below.  The index is always in [0, A).
        float best = -__builtin_huge_valf();
        int arg = 0x7FFFFFFF;
#pragma unroll
        for (int c_ = 0; c_ < CPT; ++c_) {
            const int c = sub * CPT + c_;
            if (c < A) {
                float v = 0.0f;
#pragma unroll
                for (int wv = 0; wv < NW; ++wv) v = v + Y[wv * (TM * 32) + i * 32 + c];
                v = v + b2v[c_];
                if (ok && a.q) a.q[static_cast<size_t>(r) * A + c] = v;
                if (v > best || (v != v && best == best)) {  // first maximum, like torch.argmax
                    best = v;
                    arg = c;
                }
            }
        }
#pragma unroll
        for (int d = 1; d < TPR; d <<= 1) {   // the TPR threads of a row are neighbouring lanes
            const float ob = __shfl_xor(best, d);
            const int oa = __shfl_xor(arg, d);
            const bool on = ob != ob, bn = best != best;
            if (ob > best || (on && !bn) || ((ob == best || (on && bn)) && oa < arg)) {
                best = ob;
                arg = oa;
            }
        }
        arg = static_cast<unsigned>(arg) >= static_cast<unsigned>(A) ? 0 : arg;
        if (ok && sub == 0 && a.actions) {
            if (a.explore_u) {  // u < epsilon  <=>  u * (A / epsilon) < A: that product's integer part is the uniform action
                const int k = static_cast<int>(a.explore_u[r] * a.explore_scale);
                if (static_cast<unsigned>(k) < static_cast<unsigned>(A)) arg = k;
            }
            a.actions[r] = arg;
        }
    }
    RG_ASTAMP_END(a, E, N, A, H, cb, lane);
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

// torch layout [S][3H][H] float32 -> three bfloat16 planes in the SPLIT kernel's streaming order
// [S][cb][ks][gate][plane][lane = (half, col)][8]: 6 bytes per weight
__global__ void pack_gru_bf16x3_kernel(const float *src, uint16_t *dst, int n_sets, int H) {
    const int nks = H / 16, ncb = H / 32;
    const size_t total = static_cast<size_t>(n_sets) * 3 * H * H;   // (set, cb, ks, gate, lane, e) tuples = weights
    for (size_t o = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<size_t>(gridDim.x) * blockDim.x) {
        size_t r = o;
        const int e = r % 8; r /= 8;
        const int lane = r % 64; r /= 64;
        const int g = r % 3; r /= 3;
        const int ks = r % nks; r /= nks;
        const int cb = r % ncb; r /= ncb;
        const int s = static_cast<int>(r);
        const int half = lane >> 5, col = lane & 31;
        const int row = g * H + cb * 32 + col, k = ks * 16 + half * 8 + e;
        const float w = src[(static_cast<size_t>(s) * 3 * H + row) * H + k];
        const uint32_t bh = __builtin_bit_cast(uint32_t, w) & 0xFFFF0000u;
        const float r1 = w - __builtin_bit_cast(float, bh);
        const uint32_t bm = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, bm);
        const uint32_t bl = __builtin_bit_cast(uint32_t, r2);
        const size_t base = ((((static_cast<size_t>(s) * ncb + cb) * nks + ks) * 3 + g) * 3) * 64 * 8 + static_cast<size_t>(lane) * 8 + e;
        dst[base] = static_cast<uint16_t>(bh >> 16);
        dst[base + 64 * 8] = static_cast<uint16_t>(bm >> 16);
        dst[base + 2 * 64 * 8] = static_cast<uint16_t>(bl >> 16);
    }
}

// torch layout [S][3H][H] float32 -> two binary16 planes (hi, 2^11 lo: see split1) in the streaming order
// [S][cb][ks][gate][plane][lane = (half, col)][8]: 4 bytes per weight.  Round to nearest here (the weights are split once).
__global__ void pack_gru_f16x2_kernel(const float *src, uint16_t *dst, int n_sets, int H) {
    const int nks = H / 16, ncb = H / 32;
    const size_t total = static_cast<size_t>(n_sets) * 3 * H * H;
    for (size_t o = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<size_t>(gridDim.x) * blockDim.x) {
        size_t r = o;
        const int e = r % 8; r /= 8;
        const int lane = r % 64; r /= 64;
        const int g = r % 3; r /= 3;
        const int ks = r % nks; r /= nks;
        const int cb = r % ncb; r /= ncb;
        const int s = static_cast<int>(r);
        const int half = lane >> 5, col = lane & 31;
        const int row = g * H + cb * 32 + col, k = ks * 16 + half * 8 + e;
        const float w = f16_saturate(src[(static_cast<size_t>(s) * 3 * H + row) * H + k]);
        const _Float16 hi = static_cast<_Float16>(w);
        const float rest = (w - static_cast<float>(hi)) * F16_LO_SCALE;
        const _Float16 lo = static_cast<_Float16>(rest);
        const size_t base = ((((static_cast<size_t>(s) * ncb + cb) * nks + ks) * 3 + g) * 2) * 64 * 8 + static_cast<size_t>(lane) * 8 + e;
        dst[base] = __builtin_bit_cast(uint16_t, hi);
        dst[base + 64 * 8] = __builtin_bit_cast(uint16_t, lo);
    }
}

}  // namespace rg

static thread_local char g_actor_err[256] = "";

extern "C" int rg_actor_pack_gru_f16x2(const float *src, int32_t n_sets, int32_t hidden_dim, void *dst, void *hip_stream) {
    if (!src || !dst || n_sets < 1 || (hidden_dim != 64 && hidden_dim != 128)) {
        snprintf(g_actor_err, sizeof(g_actor_err), "rg_actor_pack_gru_f16x2: NULL array, n_sets < 1 or hidden_dim not 64 / 128");
        return -1;
    }
    if (reinterpret_cast<uintptr_t>(dst) & 15u) {
        snprintf(g_actor_err, sizeof(g_actor_err), "rg_actor_pack_gru_f16x2: dst must be 16-byte aligned");
        return -9;
    }
    hipLaunchKernelGGL(rg::pack_gru_f16x2_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(hip_stream), src,
                       static_cast<uint16_t *>(dst), n_sets, hidden_dim);
    return hipGetLastError() == hipSuccess ? 0 : -30;
}

extern "C" int rg_actor_pack_gru_bf16x3(const float *src, int32_t n_sets, int32_t hidden_dim, void *dst, void *hip_stream) {
    if (!src || !dst || n_sets < 1 || (hidden_dim != 64 && hidden_dim != 128)) {
        snprintf(g_actor_err, sizeof(g_actor_err), "rg_actor_pack_gru_bf16x3: NULL array, n_sets < 1 or hidden_dim not 64 / 128");
        return -1;
    }
    if (reinterpret_cast<uintptr_t>(dst) & 15u) {
        snprintf(g_actor_err, sizeof(g_actor_err), "rg_actor_pack_gru_bf16x3: dst must be 16-byte aligned");
        return -9;
    }
    hipLaunchKernelGGL(rg::pack_gru_bf16x3_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(hip_stream), src,
                       static_cast<uint16_t *>(dst), n_sets, hidden_dim);
    return hipGetLastError() == hipSuccess ? 0 : -30;
}

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

RG_ACTOR_DIAG_ENTRY   // (diagnostic builds: rg_actor_occupancy)

static int actor_forward(const rg_actor_weights *w, int32_t num_envs, int32_t n_agents, const float *obs,
                         int32_t obs_dim, int32_t append_agent_id, const uint8_t *restart, float *hidden,
                         float *q, int32_t *actions, const float *explore_u, float epsilon, void *hip_stream) {
    auto fail = [](int code, const char *msg) {
        snprintf(g_actor_err, sizeof(g_actor_err), "%s", msg);
        return code;
    };
    if (!w || !obs || !hidden) return fail(-1, "weights, obs or hidden is NULL");
    if (!w->w1 || !w->b1 || !w->wih || !w->bih || !w->w2 || !w->b2) return fail(-2, "a weight array is NULL");
    if (w->use_rnn && (!w->whh || !w->bhh)) return fail(-2, "GRU weights whh / bhh are NULL");
    if (w->hidden_dim != 64 && w->hidden_dim != 128) return fail(-3, "hidden_dim must be 64 or 128 (the reference's actors)");
    if (w->n_actions < 1 || w->n_actions > 32) return fail(-4, "n_actions must be in 1..32");
    if (w->gru_packed < 0 || w->gru_packed > 3) return fail(-10, "gru_packed must be 0 (torch layout), 1 (rg_actor_pack_gru), 2 (rg_actor_pack_gru_bf16x3) or 3 (rg_actor_pack_gru_f16x2)");
    if (w->n_sets != 1 && w->n_sets != n_agents) return fail(-5, "n_sets must be 1 (shared) or n_agents");
    if (num_envs < 1 || n_agents < 1 || obs_dim < 1) return fail(-6, "num_envs, n_agents, obs_dim must be >= 1");
    if (explore_u && !(epsilon >= 1e-6f && epsilon <= 1.0f)) return fail(-11, "epsilon must be in [1e-6, 1] when explore_u is given");
    if (explore_u && !actions) return fail(-11, "explore_u without an actions array");
    const int in_dim = obs_dim + (append_agent_id ? n_agents : 0);
    if (in_dim != w->input_dim) return fail(-7, "obs_dim (+ n_agents with append_agent_id) != the actor's input_dim");
    const int ip = (in_dim + 7) / 8 * 8;
    if (ip > rg::MAX_IP) return fail(-8, "input_dim above 64 is not supported");
    if ((reinterpret_cast<uintptr_t>(hidden) | reinterpret_cast<uintptr_t>(w->w1) | reinterpret_cast<uintptr_t>(w->wih) |
         reinterpret_cast<uintptr_t>(w->whh) | reinterpret_cast<uintptr_t>(w->w2)) & 15u)
        return fail(-9, "hidden, w1, wih, whh and w2 must be 16-byte aligned");
    rg::ActorArgs a;
    a.w = *w;
    a.obs = obs;
    a.restart = restart;
    a.hidden = hidden;
    a.q = q;
    a.actions = actions;
    a.explore_u = explore_u;
    a.explore_scale = explore_u ? static_cast<float>(w->n_actions) / epsilon : 0.0f;
    a.E = num_envs;
    a.N = n_agents;
    a.D = obs_dim;
    a.append_agent_id = append_agent_id;
    a.ip = ip;
    const int tiles = w->n_sets == 1 ? (num_envs * n_agents + rg::TM - 1) / rg::TM
                                     : n_agents * ((num_envs + rg::TM - 1) / rg::TM);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const int split = !w->use_rnn ? 0 : w->gru_packed == 2 ? 1 : w->gru_packed == 3 ? 2 : 0;
    if (w->hidden_dim == 64) {
        if (split == 2) hipLaunchKernelGGL((rg::actor_kernel<64, 2>), dim3(tiles), dim3(128), 0, stream, a);
        else if (split == 1) hipLaunchKernelGGL((rg::actor_kernel<64, 1>), dim3(tiles), dim3(128), 0, stream, a);
        else hipLaunchKernelGGL((rg::actor_kernel<64, 0>), dim3(tiles), dim3(128), 0, stream, a);
    } else {
        if (split == 2) hipLaunchKernelGGL((rg::actor_kernel<128, 2>), dim3(tiles), dim3(256), 0, stream, a);
        else if (split == 1) hipLaunchKernelGGL((rg::actor_kernel<128, 1>), dim3(tiles), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((rg::actor_kernel<128, 0>), dim3(tiles), dim3(256), 0, stream, a);
    }
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(-30, hipGetErrorString(err));
    return 0;
}

extern "C" int rg_actor_forward(const rg_actor_weights *w, int32_t num_envs, int32_t n_agents, const float *obs,
                                int32_t obs_dim, int32_t append_agent_id, const uint8_t *restart, float *hidden,
                                float *q, int32_t *actions, void *hip_stream) {
    return actor_forward(w, num_envs, n_agents, obs, obs_dim, append_agent_id, restart, hidden, q, actions, nullptr, 0.0f,
                         hip_stream);
}

extern "C" int rg_actor_forward_explore(const rg_actor_weights *w, int32_t num_envs, int32_t n_agents, const float *obs,
                                        int32_t obs_dim, int32_t append_agent_id, const uint8_t *restart, float *hidden,
                                        float *q, int32_t *actions, const float *explore_u, float epsilon,
                                        void *hip_stream) {
    return actor_forward(w, num_envs, n_agents, obs, obs_dim, append_agent_id, restart, hidden, q, actions, explore_u, epsilon,
                         hip_stream);
}
