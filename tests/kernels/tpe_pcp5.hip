// One instantiation of the thread-per-env step kernel (PredatorCapturePrey, N = 5, single step) for
// tests/test_kernel_resources.py: compiled alone in a few seconds to read the compiler's resource report.
#include "step_tpe.h"
namespace rg {
template __global__ void tpe::step_kernel<RG_SCN_PREDATOR_CAPTURE_PREY, 5, false>(const KernelArgs);
template __global__ void tpe::step_kernel<RG_SCN_PREDATOR_CAPTURE_PREY, 6, false>(const KernelArgs);
template __global__ void tpe::step_kernel<RG_SCN_PREDATOR_CAPTURE_PREY, 4, false>(const KernelArgs);
}
