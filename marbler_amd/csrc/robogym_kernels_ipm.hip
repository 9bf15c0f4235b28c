// robogym_kernels_ipm.hip -- instantiates the lane-group kernels of the interior-point mode (step_group.h, QPM = RG_QP_CVXOPT) for
// one env step per launch.  Its own translation unit: these kernels are long chains of binary64 arithmetic in which the max-ILP
// scheduling strategy pays (build.py FILE_FLAGS; 4096 x 5: 142 -> 134 us, Warehouse 4096 x 8: 607 -> 548), while the
// exact-projection kernels are better off with the default one.
#include "step_group.h"

namespace rg {

hipError_t launch_step_ipm(const KernelArgs &a, int grid, hipStream_t stream) { return launch_ipm_group<false>(a, grid, stream); }

}  // namespace rg
