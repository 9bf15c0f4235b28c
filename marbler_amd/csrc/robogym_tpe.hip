// robogym_tpe.hip -- instantiates the thread-per-env kernels (step_tpe.h) for one env step per launch.
#include "step_tpe.h"
#include "step_tpe_ipm.h"

namespace rg {

bool tpe_supported(const rg_scenario_params &p) {
    if (p.qp_mode == RG_QP_CVXOPT && p.n_agents > 5) return false;   // that mode's rows + KKT matrix fit one lane's registers up to N = 5
    if (p.scenario == RG_SCN_ARCTIC_TRANSPORT) return p.n_agents == 4;
    if (p.scenario == RG_SCN_MATERIAL_TRANSPORT && p.n_agents < 4) return false;
    return p.n_agents >= 2 && p.n_agents <= 6;   // N >= 7: the lane-group kernel at every batch size (step_tpe.h launch_scn)
}

hipError_t launch_step_tpe(const KernelArgs &a, hipStream_t stream) { return launch_tpe<false>(a, stream); }

}  // namespace rg
