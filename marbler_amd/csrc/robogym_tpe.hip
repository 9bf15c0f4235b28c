// robogym_tpe.hip -- instantiates the thread-per-env kernels (step_tpe.h) for one env step per launch.
#include "step_tpe.h"

namespace rg {

bool tpe_supported(const rg_scenario_params &p) {
    if (p.qp_mode != RG_QP_EXACT) return false;   // the interior-point mode runs on the lane-group kernel (step_group.h)
    if (p.scenario == RG_SCN_ARCTIC_TRANSPORT) return p.n_agents == 4;
    if (p.scenario == RG_SCN_MATERIAL_TRANSPORT && p.n_agents < 4) return false;
    return p.n_agents >= 2 && p.n_agents <= 6;   // N >= 7: the lane-group kernel at every batch size (step_tpe.h launch_scn)
}

hipError_t launch_step_tpe(const KernelArgs &a, hipStream_t stream) { return launch_tpe<false>(a, stream); }

}  // namespace rg
