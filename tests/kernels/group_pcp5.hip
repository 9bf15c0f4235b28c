// One instantiation each of the lane-group step kernel for tests/test_kernel_resources.py (see tpe_pcp5.hip).
#include "step_group.h"
namespace rg {
template __global__ void step_kernel<RG_SCN_PREDATOR_CAPTURE_PREY, 8, false, 5, false>(const KernelArgs);
template __global__ void step_kernel<RG_SCN_WAREHOUSE, 8, false, 8, false>(const KernelArgs);
template __global__ void step_kernel<RG_SCN_MATERIAL_TRANSPORT, 8, false, 6, false>(const KernelArgs);
}
