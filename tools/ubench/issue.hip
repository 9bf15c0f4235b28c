// Issue-rate microbenchmark for ONE wavefront on a SIMD of gfx950 (diagnostic, not shipped):
// core cycles per VALU instruction for dependent / independent chains, partial EXEC masks,
// DPP-fed chains, transcendental and compare->select chains.  Each loop body is one asm block
// (the compiler adds no hazard nops inside it).  Driver: tools/ubench/issue.py.
#include <hip/hip_runtime.h>

constexpr int REP = 128;  // loop trips; 16 instructions per trip

#define X4(a) a a a a
#define X16(a) X4(a) X4(a) X4(a) X4(a)
#define X8(a) X4(a) X4(a)

template <int MODE>
__global__ void bench(float *out, long long *cyc, int active) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    const float m = 0.999f, c = 1e-3f;
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < active) {
        t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)");
        for (int i = 0; i < REP; ++i) {
            if constexpr (MODE == 0)
                asm volatile(X16("v_fma_f32 %0, %0, %4, %5\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
            else if constexpr (MODE == 1)
                asm volatile(X8("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\t")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
            else if constexpr (MODE == 2)
                asm volatile(X4("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
            else if constexpr (MODE == 3)  // DPP move of the fresh result, then fma on it (the pair-round pattern)
                asm volatile(X8("s_nop 1\n\tv_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_fma_f32 %0, %1, %4, %0\n\t")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
            else if constexpr (MODE == 4)
                asm volatile(X16("v_rcp_f32 %0, %0\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
            else if constexpr (MODE == 5)
                asm volatile(X8("v_cmp_gt_f32 vcc, %0, %5\n\tv_cndmask_b32 %0, %0, %4, vcc\n\t")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c) : "vcc");
            else if constexpr (MODE == 6)  // dependent chain with an independent instruction after every link
                asm volatile(X8("v_fma_f32 %0, %0, %4, %5\n\tv_mul_f32 %1, %4, %5\n\t")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
            else if constexpr (MODE == 7)  // dependent chain, 3 independent fillers per link
                asm volatile(X4("v_fma_f32 %0, %0, %4, %5\n\tv_mul_f32 %1, %4, %5\n\tv_mul_f32 %2, %4, %5\n\tv_mul_f32 %3, %4, %5\n\t")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

extern "C" int run_mode(int mode, int active, float *out, long long *cyc) {
    switch (mode) {
        case 0: bench<0><<<1, 64>>>(out, cyc, active); break;
        case 1: bench<1><<<1, 64>>>(out, cyc, active); break;
        case 2: bench<2><<<1, 64>>>(out, cyc, active); break;
        case 3: bench<3><<<1, 64>>>(out, cyc, active); break;
        case 4: bench<4><<<1, 64>>>(out, cyc, active); break;
        case 5: bench<5><<<1, 64>>>(out, cyc, active); break;
        case 6: bench<6><<<1, 64>>>(out, cyc, active); break;
        case 7: bench<7><<<1, 64>>>(out, cyc, active); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ packed f32 against scalar f32 (round 5)
// Does `v_pk_fma_f32` (two binary32 fmas per lane per instruction, even-aligned register pairs) issue faster than the two
// `v_fma_f32` it replaces?  INDEPENDENT streams (8 accumulator pairs per wave, no instruction depends on the one before it),
// 1 / 2 / 4 waves per SIMD (one workgroup of 4 / 8 / 16 waves on one CU), every wave timed with s_memtime; the driver
// (issue.py --pk) reports flop / clk / SIMD = waves_per_simd x instructions x 64 lanes x flops per lane-instruction / cycles.
// MODE 0: 16 x v_fma_f32 per trip (2 flop per lane each);  MODE 1: 8 x v_pk_fma_f32 per trip (4 flop per lane each): the
// SAME arithmetic per trip;  MODE 2 / 3: v_mul_f32 + v_add_f32 pairs against v_pk_mul_f32 + v_pk_add_f32 (the unfused form
// the -ffp-contract=off step kernels consist of).
typedef float pkf2 __attribute__((ext_vector_type(2)));
constexpr int PK_REP = 256;
template <int MODE>
__global__ __launch_bounds__(1024) void bench_pk(float *out, long long *cyc) {
    pkf2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = pkf2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
    const pkf2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < PK_REP; ++i) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_fma_f32 %0, %0, %2, %4\n\tv_fma_f32 %1, %1, %3, %5" : "+v"(a[j].x), "+v"(a[j].y) : "v"(m.x), "v"(m.y), "v"(c.x), "v"(c.y));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(c));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5"
                             : "+v"(a[j].x), "+v"(a[j].y) : "v"(m.x), "v"(m.y), "v"(c.x), "v"(c.y));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(a[j]) : "v"(m), "v"(c));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// waves_per_simd in {1, 2, 4}: one workgroup of 4 x that many waves (a CU has 4 SIMDs); cyc[w] = ticks of wave w
extern "C" int run_pk(int mode, int waves_per_simd, float *out, long long *cyc) {
    const int threads = 256 * waves_per_simd;
    switch (mode) {
        case 0: bench_pk<0><<<1, threads>>>(out, cyc); break;
        case 1: bench_pk<1><<<1, threads>>>(out, cyc); break;
        case 2: bench_pk<2><<<1, threads>>>(out, cyc); break;
        case 3: bench_pk<3><<<1, threads>>>(out, cyc); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
