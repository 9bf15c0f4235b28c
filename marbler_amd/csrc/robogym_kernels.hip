// robogym_kernels.hip -- instantiates the lane-group kernels (step_group.h) for one env step per launch
// (rg_step, rg_get_obs) and the explicit reset (rg_reset).
#include "step_group.h"

namespace rg {

template <int SCN>
static hipError_t launch_reset_scn(const KernelArgs &a, hipStream_t stream) {
    const int gw = group_width(a.p.n_agents);
    const int grid = (a.E + WAVE / gw - 1) / (WAVE / gw);
    if (gw == 4) hipLaunchKernelGGL((reset_kernel<SCN, 4>), dim3(grid), dim3(WAVE), 0, stream, a);
    else if (gw == 8) hipLaunchKernelGGL((reset_kernel<SCN, 8>), dim3(grid), dim3(WAVE), 0, stream, a);
    else hipLaunchKernelGGL((reset_kernel<SCN, 16>), dim3(grid), dim3(WAVE), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_step(const KernelArgs &a, bool obs_only, hipStream_t stream) {
    return obs_only ? launch_step_group<true, false>(a, stream) : launch_step_group<false, false>(a, stream);
}

hipError_t launch_reset(const KernelArgs &a, hipStream_t stream) {
    switch (a.p.scenario) {
        case RG_SCN_PREDATOR_CAPTURE_PREY:
            return launch_reset_scn<RG_SCN_PREDATOR_CAPTURE_PREY>(a, stream);
        case RG_SCN_WAREHOUSE:
            return launch_reset_scn<RG_SCN_WAREHOUSE>(a, stream);
        case RG_SCN_MATERIAL_TRANSPORT:
            return launch_reset_scn<RG_SCN_MATERIAL_TRANSPORT>(a, stream);
        case RG_SCN_SIMPLE:
            return launch_reset_scn<RG_SCN_SIMPLE>(a, stream);
        case RG_SCN_ARCTIC_TRANSPORT:
            hipLaunchKernelGGL((reset_kernel<RG_SCN_ARCTIC_TRANSPORT, 4>), dim3((a.E + 15) / 16), dim3(WAVE), 0, stream, a);
            return hipGetLastError();
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace rg
