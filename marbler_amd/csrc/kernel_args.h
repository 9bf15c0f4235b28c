// kernel_args.h -- the one argument block of every launch (host <-> device contract).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/robogym.h"

namespace rg {

struct KernelArgs {
    rg_scenario_params p;
    rg_state st;
    rg_step_io io;
    const int32_t *actions;
    const uint8_t *reset_mask;
    int32_t E;
    int32_t auto_reset;
    int64_t env_offset;
    uint64_t seed;
};

hipError_t launch_step(const KernelArgs &a, bool obs_only, hipStream_t stream);
hipError_t launch_reset(const KernelArgs &a, hipStream_t stream);

}  // namespace rg
