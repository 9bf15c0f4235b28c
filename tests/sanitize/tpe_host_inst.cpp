// Host simulation of the thread-per-env step kernel: ONE agent count per translation unit (-DSIM_N=2..8), every scenario,
// rg_step form and (-DSIM_ROLLOUT) rg_rollout form.  step_tpe.h and everything it includes are the shipped device headers;
// <hip/hip_runtime.h> resolves to tests/sanitize/hip_shim/.  See tpe_host.cpp.
#include "step_tpe.h"

#ifndef SIM_N
#error "-DSIM_N=<agents>"
#endif
#define SIM_CAT2(a, b) a##b
#define SIM_CAT(a, b) SIM_CAT2(a, b)

namespace {
template <int SCN, bool ROLLOUT>
void run(const rg::KernelArgs &a) {
    const int grid = (a.E + rg::WAVE - 1) / rg::WAVE;
    hipLaunchKernelGGL((rg::tpe::step_kernel<SCN, SIM_N, ROLLOUT>), dim3(grid), dim3(rg::WAVE), 0, nullptr, a);
}
template <bool ROLLOUT>
int dispatch(const rg::KernelArgs &a) {
    switch (a.p.scenario) {
        case RG_SCN_PREDATOR_CAPTURE_PREY: run<RG_SCN_PREDATOR_CAPTURE_PREY, ROLLOUT>(a); return 0;
        case RG_SCN_WAREHOUSE: run<RG_SCN_WAREHOUSE, ROLLOUT>(a); return 0;
        case RG_SCN_SIMPLE: run<RG_SCN_SIMPLE, ROLLOUT>(a); return 0;
#if SIM_N >= 4
        case RG_SCN_MATERIAL_TRANSPORT: run<RG_SCN_MATERIAL_TRANSPORT, ROLLOUT>(a); return 0;
#endif
#if SIM_N == 4
        case RG_SCN_ARCTIC_TRANSPORT: run<RG_SCN_ARCTIC_TRANSPORT, ROLLOUT>(a); return 0;
#endif
        default: return -1;
    }
}
}  // namespace

// rollout != 0: a.num_steps env steps in the one launch (the ROLLOUT instantiation)
int SIM_CAT(sim_step_tpe_n, SIM_N)(const rg::KernelArgs &a, int rollout) {
#ifdef SIM_ROLLOUT
    if (rollout) return dispatch<true>(a);
#else
    if (rollout) return -2;
#endif
    return dispatch<false>(a);
}
