// Does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=N: the CP puts the first kernel arguments into SGPRs while it
// sets the wave up) shorten a short launch whose first action is a load through a pointer argument?  The step kernels read
// ~30 pointers from a by-value struct; only leading plain arguments can be preloaded.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off k.hip -o plain
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=4 k.hip -o preload
// Measured (NOTEBOOK.md round 4, 5e): 11.28 us plain, 11.46 us with preloading.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
struct Big { const float *p[12]; int n[40]; };
extern "C" __global__ __launch_bounds__(64) void probe(const float *src, float *out, Big b) {
    const int i = threadIdx.x + blockIdx.x * 64;
    float v = src[i];                       // first dependent trip: needs `src`
    v += b.p[3][threadIdx.x] * (float)b.n[7];   // second: needs the struct
    for (int k = 0; k < 2500; ++k) v = v * 1.0001f + 0.5f;   // a dependent chain of ~10 us: the launch is GPU-bound
    out[i] = v;
}
int main(int argc, char **argv) {
    const int waves = 1024, n = waves * 64, reps = 3000;
    float *src, *out, *aux;
    hipMalloc(&src, n * 4); hipMalloc(&out, n * 4); hipMalloc(&aux, 4096);
    hipMemset(src, 0, n * 4); hipMemset(aux, 0, 4096);
    Big b; for (auto &p : b.p) p = aux; for (auto &x : b.n) x = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 5; ++round) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, src, out, b);
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, src, out, b);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f us per launch\n", argv[0], ms * 1e3 / reps);
    }
    return 0;
}
