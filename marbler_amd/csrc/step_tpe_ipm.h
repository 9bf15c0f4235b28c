// step_tpe_ipm.h -- the thread-per-env kernel's side of the interior-point mode (step_tpe.h ipm_lane): included by the
// instantiation units (robogym_tpe.hip, robogym_rollout_tpe.hip) only.
#pragma once
#include "ipm_qp.h"
#include "step_tpe.h"

namespace rg {
namespace tpe {

template <int N>
__device__ int ipm_lane(const rg_scenario_params &p, const float (&xix)[N], const float (&xiy)[N], float (&ux)[N], float (&uy)[N],
                        float *rec) {
    float4 *r4 = reinterpret_cast<float4 *>(rec);
#pragma unroll
    for (int a = 0; a < N; ++a) r4[a] = make_float4(xix[a], xiy[a], ux[a], uy[a]);
    stage_fence();
    const int iters = ipm::solve_qp<N, 1>(ipm::make_consts(p), r4, nullptr, 0);
    stage_fence();
#pragma unroll
    for (int a = 0; a < N; ++a) {
        const float4 r = r4[a];
        ux[a] = r.z;
        uy[a] = r.w;
    }
    return iters;
}

}  // namespace tpe
}  // namespace rg
