// TEST BUILD ONLY (tests/test_sanitizers.py): the C ABI's host half (marbler_amd/csrc/robogym_capi.hip: parameter
// validation, handle bookkeeping, error reporting) compiled for the host alone with ASan + UBSan.  The kernel
// launchers live in the device translation units, which a host-only sanitizer build does not contain; these
// definitions satisfy the linker and refuse to run -- the no-GPU tests never reach a launch (rg_create fails first).
#include "kernel_args.h"

namespace rg {
hipError_t launch_step(const KernelArgs &, bool, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_reset(const KernelArgs &, hipStream_t) { return hipErrorNotSupported; }
bool tpe_supported(const rg_scenario_params &) { return false; }
hipError_t launch_step_tpe(const KernelArgs &, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_rollout(const KernelArgs &, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_rollout_tpe(const KernelArgs &, hipStream_t) { return hipErrorNotSupported; }
}  // namespace rg

extern "C" {
int rg_actor_forward(const rg_actor_weights *, int32_t, int32_t, const float *, int32_t, int32_t, const uint8_t *, float *,
                     float *, int32_t *, void *) { return -100; }
int rg_actor_forward_explore(const rg_actor_weights *, int32_t, int32_t, const float *, int32_t, int32_t, const uint8_t *,
                             float *, float *, int32_t *, const float *, float, void *) { return -100; }
int rg_actor_pack_gru(const float *, int32_t, int32_t, float *, void *) { return -100; }
int rg_actor_pack_gru_f16x2(const float *, int32_t, int32_t, void *, void *) { return -100; }
int rg_actor_pack_gru_bf16x3(const float *, int32_t, int32_t, void *, void *) { return -100; }
const char *rg_actor_last_error(void) { return "host-only sanitizer build: no kernels"; }
}
