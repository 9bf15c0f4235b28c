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
