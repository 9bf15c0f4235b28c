// robogym_rollout_tpe_hi.hip -- the N = 7, 8 instantiations of the thread-per-env kernels (step_tpe.h) for rg_rollout.
#define RG_TPE_HI
#include "step_tpe.h"

namespace rg {

hipError_t launch_rollout_tpe_hi(const KernelArgs &a, hipStream_t stream) { return launch_tpe<true>(a, stream); }

}  // namespace rg
