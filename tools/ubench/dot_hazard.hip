// dot_hazard.hip -- minimal reproducer of the DOT -> VALU data hazard on gfx950.
// out[lane] = min(dot2(d, d), big) computed three ways: the consumer 0, 1, 2 and 3 wait states behind the v_dot2.  With fewer
// than three the v_min reads the register's OLD contents (here: the packed binary16 input itself); the hardware does not
// interlock and the assembler does not warn.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/ubench/libdot_hazard.so tools/ubench/dot_hazard.hip
#include <hip/hip_runtime.h>

template <int NOPS>
__global__ void dot_then_min(const int *in, int *out) {
    int d = in[threadIdx.x];
    const int big = 0x7f000000;
    int r;
    if constexpr (NOPS == 0) asm volatile("v_dot2_f32_f16 %0, %1, %1, 0\n\tv_min_i32 %0, %0, %2" : "=&v"(r) : "v"(d), "v"(big));
    if constexpr (NOPS == 1) asm volatile("v_dot2_f32_f16 %0, %1, %1, 0\n\ts_nop 0\n\tv_min_i32 %0, %0, %2" : "=&v"(r) : "v"(d), "v"(big));
    if constexpr (NOPS == 2) asm volatile("v_dot2_f32_f16 %0, %1, %1, 0\n\ts_nop 1\n\tv_min_i32 %0, %0, %2" : "=&v"(r) : "v"(d), "v"(big));
    if constexpr (NOPS == 3) asm volatile("v_dot2_f32_f16 %0, %1, %1, 0\n\ts_nop 2\n\tv_min_i32 %0, %0, %2" : "=&v"(r) : "v"(d), "v"(big));
    out[threadIdx.x] = r;
}
// the in-place form the kernels used (destination = source register): the stale value IS the input
template <int NOPS>
__global__ void dot_in_place(const int *in, int *out) {
    int d = in[threadIdx.x];
    const int big = 0x7f000000;
    if constexpr (NOPS == 0) asm volatile("v_dot2_f32_f16 %0, %0, %0, 0\n\tv_min_i32 %0, %0, %1" : "+v"(d) : "v"(big));
    if constexpr (NOPS == 3) asm volatile("v_dot2_f32_f16 %0, %0, %0, 0\n\ts_nop 2\n\tv_min_i32 %0, %0, %1" : "+v"(d) : "v"(big));
    out[threadIdx.x] = d;
}

extern "C" void run(int variant, const int *in, int *out, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (variant) {
        case 0: hipLaunchKernelGGL(dot_then_min<0>, dim3(1), dim3(64), 0, s, in, out); break;
        case 1: hipLaunchKernelGGL(dot_then_min<1>, dim3(1), dim3(64), 0, s, in, out); break;
        case 2: hipLaunchKernelGGL(dot_then_min<2>, dim3(1), dim3(64), 0, s, in, out); break;
        case 3: hipLaunchKernelGGL(dot_then_min<3>, dim3(1), dim3(64), 0, s, in, out); break;
        case 10: hipLaunchKernelGGL(dot_in_place<0>, dim3(1), dim3(64), 0, s, in, out); break;
        case 13: hipLaunchKernelGGL(dot_in_place<3>, dim3(1), dim3(64), 0, s, in, out); break;
    }
}
