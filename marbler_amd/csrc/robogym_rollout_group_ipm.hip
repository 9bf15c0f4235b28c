// robogym_rollout_group_ipm.hip -- the lane-group kernels of the interior-point mode for rg_rollout (see robogym_kernels_ipm.hip).
#include "step_group.h"

namespace rg {

hipError_t launch_rollout_ipm(const KernelArgs &a, int grid, hipStream_t stream) { return launch_ipm_group<true>(a, grid, stream); }

}  // namespace rg
