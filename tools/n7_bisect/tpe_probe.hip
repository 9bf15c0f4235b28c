// tpe_probe.hip -- ONE instantiation of the thread-per-env step kernel (step_tpe.h) behind a tiny C entry point, so that the
// same source can be compiled many times with different compiler flags / -opt-bisect-limit values and all the variants
// tested on the GPU in one call (tools/n7_bisect/run_probe.py).  Round-4 investigation of the MaterialTransport N = 7
// miscompute under -O3 -fno-slp-vectorize (DESIGN.md / NOTEBOOK.md); not part of the library.
#include <string.h>
#include "step_tpe.h"

#ifndef PROBE_SCN
#define PROBE_SCN RG_SCN_MATERIAL_TRANSPORT
#endif
#ifndef PROBE_N
#define PROBE_N 7
#endif

extern "C" int probe_step(const rg_scenario_params *p, const rg_state *st, const rg_step_io *io, const int32_t *actions, int32_t E,
                          int32_t auto_reset, uint64_t seed, int64_t env_offset, void *stream) {
    if (p->scenario != PROBE_SCN || p->n_agents != PROBE_N) return -1;
    rg::KernelArgs a;
    memset(&a, 0, sizeof(a));
    a.p = *p;
    a.k = rg::make_consts(*p);
    a.st = *st;
    a.st.next_init = nullptr;
    a.st.next_episode = nullptr;
    a.io = *io;
    a.actions = actions;
    a.E = E;
    a.num_steps = 1;
    a.auto_reset = auto_reset;
    a.seed = seed;
    a.env_offset = env_offset;
    const int grid = (E + rg::WAVE - 1) / rg::WAVE;
    hipLaunchKernelGGL((rg::tpe::step_kernel<PROBE_SCN, PROBE_N, false>), dim3(grid), dim3(rg::WAVE), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}
