// HOST SIMULATION SHIM (test infrastructure, tests/test_sanitizers.py) -- stands in for <hip/hip_runtime.h> so that the
// thread-per-env step kernel (marbler_amd/csrc/step_tpe.h, with device_common.h / sim_math.h / kernel_args.h exactly as
// shipped) compiles as plain host C++ and runs under ASan + UBSan and under MSan, which do not exist for gfx950 on this
// pool.  A workgroup of 64 lanes is 64 host threads; everything the wavefront's lock step guarantees on the GPU is made
// explicit here:
//   * __syncthreads(), the wave-scope fence of stage_fence() and every cross-lane builtin (ballot, readlane,
//     readfirstlane) are barriers over the 64 threads (plus a shared exchange slot);
//   * `__shared__` is a function-local static (one workgroup runs at a time);
//   * DPP permutes do not occur in this kernel (they abort if reached).
// The arithmetic is the same IEEE binary32 sequence (-ffp-contract=off; explicit fma), so the results must equal the float32
// oracle bit for bit -- the harness (tpe_host.cpp) checks that while the sanitizers watch every load, store, shift,
// conversion and branch on an uninitialised value.
#pragma once
#ifndef RG_HOST_SIM
#error "hip_shim is for the host-simulation build only (-DRG_HOST_SIM)"
#endif
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
inline hipError_t hipGetLastError() { return hipSuccess; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(8) float2 {
    float x, y;
};
struct alignas(16) float4 {
    float x, y, z, w;
};
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }

namespace rg_sim {
constexpr int LANES = 64;
struct Idx {
    unsigned x, y, z;
};
extern thread_local Idx t_thread, t_block, t_grid;
extern pthread_barrier_t g_barrier;
extern volatile uint64_t g_slot[LANES];
inline void barrier() { pthread_barrier_wait(&g_barrier); }
// value exchange over the wave: every lane publishes, then reads (two barriers: the slot is reused)
inline uint64_t exchange_read(uint64_t mine, int from) {
    g_slot[t_thread.x] = mine;
    barrier();
    const uint64_t v = g_slot[from];
    barrier();
    return v;
}
inline unsigned long long ballot(bool pred) {
    g_slot[t_thread.x] = pred ? 1 : 0;
    barrier();
    unsigned long long m = 0;
    for (int i = 0; i < LANES; ++i) m |= static_cast<unsigned long long>(g_slot[i] & 1) << i;
    barrier();
    return m;
}
// runs `grid` workgroups of 64 lanes, one after the other, each lane a thread
void launch_blocks(unsigned grid, void (*lane_fn)(void *), void *arg);
template <typename K, typename A>
struct Call {
    K k;
    const A *a;
    static void run(void *p) {
        Call *c = static_cast<Call *>(p);
        c->k(*c->a);
    }
};
template <typename K, typename A>
inline void launch(K k, dim3 grid, dim3 block, const A &a) {
    if (block.x != LANES) abort();
    Call<K, A> c{k, &a};
    launch_blocks(grid.x, &Call<K, A>::run, &c);
}
}  // namespace rg_sim

#define threadIdx (rg_sim::t_thread)
#define blockIdx (rg_sim::t_block)
#define gridDim (rg_sim::t_grid)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) rg_sim::launch(kernel, grid, block, __VA_ARGS__)

inline void __syncthreads() { rg_sim::barrier(); }
inline unsigned long long __ballot(bool pred) { return rg_sim::ballot(pred); }

// ---- gfx950 builtins the kernel uses, as host functions (x86 clang does not know these names)
#define __builtin_amdgcn_fence(order, scope) rg_sim::barrier()
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }  // v_rcp_f32 is within 1 ulp; its uses tolerate that (copy_runs)
inline int __builtin_amdgcn_readfirstlane(int v) { return static_cast<int>(rg_sim::exchange_read(static_cast<uint32_t>(v), 0)); }
inline int __builtin_amdgcn_readlane(int v, int lane) { return static_cast<int>(rg_sim::exchange_read(static_cast<uint32_t>(v), lane)); }
inline unsigned long long __builtin_amdgcn_s_memtime() { return 0; }
inline int __builtin_amdgcn_update_dpp(int, int, int, int, int, bool) { abort(); }  // lane-group kernel only

// v_cvt_pkrtz_f16_f32: two floats to binary16, rounded toward zero
typedef _Float16 rg_sim_half2 __attribute__((ext_vector_type(2)));
inline uint16_t rg_sim_f32_to_f16_rtz(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int32_t ex = static_cast<int32_t>((u >> 23) & 0xFF) - 127 + 15;
    const uint32_t man = u & 0x7FFFFFu;
    if (((u >> 23) & 0xFF) == 0xFF) return static_cast<uint16_t>(sign | 0x7C00u | (man ? 0x200u : 0));
    if (ex >= 31) return static_cast<uint16_t>(sign | 0x7BFFu);  // toward zero: the largest finite value
    if (ex <= 0) {
        if (ex < -10) return static_cast<uint16_t>(sign);
        const uint32_t m = (man | 0x800000u) >> (14 - ex);
        return static_cast<uint16_t>(sign | m);
    }
    return static_cast<uint16_t>(sign | (static_cast<uint32_t>(ex) << 10) | (man >> 13));
}
inline rg_sim_half2 __builtin_amdgcn_cvt_pkrtz(float a, float b) {
    const uint16_t h[2] = {rg_sim_f32_to_f16_rtz(a), rg_sim_f32_to_f16_rtz(b)};
    rg_sim_half2 r;
    memcpy(&r, h, 4);
    return r;
}
// v_dot2_f32_f16 d, a, a, 0 on the bits of a packed binary16 pair: x*x + y*y in binary32 (the products of two binary16
// values are exact in binary32; one rounding in the sum)
inline int rg_sim_dot2_self(int bits) {
    rg_sim_half2 h;
    memcpy(&h, &bits, 4);
    const float x = static_cast<float>(h.x), y = static_cast<float>(h.y);
    const float r = x * x + y * y;
    int o;
    memcpy(&o, &r, 4);
    return o;
}
