// robogym_rollout_group.hip -- instantiates the lane-group kernels (step_group.h) for rg_rollout:
// num_steps env steps per launch.  Its own translation unit so the builds run in parallel.
#include "step_group.h"

namespace rg {

hipError_t launch_rollout(const KernelArgs &a, hipStream_t stream) { return launch_step_group<false, true>(a, stream); }

}  // namespace rg
