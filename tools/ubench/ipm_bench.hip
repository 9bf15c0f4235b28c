// ipm_bench.hip -- csrc/ipm_qp.h alone (diagnostic, not shipped): one QP per LANE GROUP of 8 (the lane-group kernel's usage: every
// lane of a group runs the same iteration on its env's LDS records) or one per LANE (the thread-per-env usage), timed with
// s_memtime per wave.  Driver: tests/ipm_bench.py (under tests/ because it checks against the CPU oracle) (instances drawn like tests/test_ipm_spec.py, results checked against
// the CPU twin bit for bit).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I marbler_amd/csrc -shared -fPIC -o tools/ubench/libipm_bench.so tools/ubench/ipm_bench.hip
#include "ipm_qp.h"

// per_lane = 0: instance g of the wave in lanes 8g .. 8g+7 (8 instances per wave); 1: instance per lane (64 per wave)
__global__ __launch_bounds__(64) void ipm_kernel(rg_scenario_params p, float4 *io, int *iters, long long *ticks, int N, int n_inst, int per_lane) {
    __shared__ float4 buf[64][8];
    __shared__ __attribute__((aligned(16))) double ws_group[8][rg::ipm::ws_doubles(8)];   // per_lane = 0: one workspace per group of 8 lanes
    const int lane = threadIdx.x;
    const int slot = per_lane ? lane : (lane >> 3);
    const int per_wave = per_lane ? 64 : 8;
    const int inst = blockIdx.x * per_wave + slot;
    const bool ok = inst < n_inst;
    if (ok && (per_lane || (lane & 7) == 0))
        for (int a = 0; a < N; ++a) buf[slot][a] = io[static_cast<size_t>(inst) * 8 + a];
    __syncthreads();
    const rg::ipm::Consts c = rg::ipm::make_consts(p);
    const long long t0 = __builtin_amdgcn_s_memtime();
    int it = 0;
    if (ok) {
        if (per_lane) it = rg::ipm::solve_qp_n<1>(N, c, buf[slot], nullptr, 0);
        else if (N <= 4) it = rg::ipm::solve_qp_n<4>(N, c, buf[slot], nullptr, lane & 3);
        else it = rg::ipm::solve_qp_n<8>(N, c, buf[slot], ws_group[slot], lane & 7);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (ok && (per_lane || (lane & 7) == 0)) {
        for (int a = 0; a < N; ++a) io[static_cast<size_t>(inst) * 8 + a] = buf[slot][a];
        iters[inst] = it;
    }
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

extern "C" int ipm_run(const rg_scenario_params *p, float4 *io, int *iters, long long *ticks, int N, int n_inst, int per_lane) {
    const int per_wave = per_lane ? 64 : 8;
    ipm_kernel<<<(n_inst + per_wave - 1) / per_wave, 64>>>(*p, io, iters, ticks, N, n_inst, per_lane);
    return static_cast<int>(hipDeviceSynchronize());
}

#ifdef RG_IPM_STAMPS
extern "C" int ipm_read_ticks(unsigned long long *out, int reset) {
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rg::ipm::ipm_ticks), sizeof(zero)) != hipSuccess) return 1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(rg::ipm::ipm_ticks), zero, sizeof(zero)) != hipSuccess) return 2;
    return 0;
}
#endif
