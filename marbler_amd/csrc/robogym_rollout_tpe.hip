// robogym_rollout_tpe.hip -- instantiates the thread-per-env kernels (step_tpe.h) for rg_rollout.
#include "step_tpe.h"
#include "step_tpe_ipm.h"

namespace rg {

hipError_t launch_rollout_tpe(const KernelArgs &a, hipStream_t stream) { return launch_tpe<true>(a, stream); }

}  // namespace rg
