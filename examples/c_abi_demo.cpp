// c_abi_demo.cpp -- the C ABI of include/robogym.h driven from a plain HIP host program: no Python, no
// torch.  What a C/C++ trainer does to use the step engine: allocate the state and output arrays,
// rg_create + rg_bind_state, rg_reset, then rg_step (or rg_rollout) per batch of actions.
//
//   python -m marbler_amd.params PredatorCapturePrey params.bin predator=3 capture=2     # the YAML -> rg_scenario_params
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/c_abi_demo.cpp -Lmarbler_amd -lrobogym_hip -Wl,-rpath,$PWD/marbler_amd -o c_abi_demo
//   ./c_abi_demo params.bin 4096 500
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "robogym.h"

#define HIP_OK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            return 2;                                                                  \
        }                                                                              \
    } while (0)
#define RG_OK(x)                                                              \
    do {                                                                      \
        if ((x) != 0) {                                                       \
            fprintf(stderr, "robogym error at line %d: %s\n", __LINE__, rg_last_error()); \
            return 3;                                                         \
        }                                                                     \
    } while (0)

template <typename T>
static T *dev_zeros(size_t n) {
    T *p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, n * sizeof(T)) != hipSuccess) return nullptr;
    return p;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s params.bin [num_envs] [steps]\n", argv[0]);
        return 1;
    }
    const int E = argc > 2 ? atoi(argv[2]) : 1024, steps = argc > 3 ? atoi(argv[3]) : 200;
    rg_scenario_params p;
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(&p, 1, sizeof(p), f) != sizeof(p) || (int)sizeof(p) != rg_sizeof_params() ||
        rg_abi_version() != RG_ABI_VERSION) {
        fprintf(stderr, "cannot read %s as an rg_scenario_params of ABI %d\n", argv[1], RG_ABI_VERSION);
        return 1;
    }
    fclose(f);
    const int N = p.n_agents, D = p.obs_dim, P = p.num_prey > 0 ? p.num_prey : 1;
    const size_t EN = (size_t)E * N;

    rg_state st;
    memset(&st, 0, sizeof(st));
    st.poses = dev_zeros<float>(EN * 3);
    st.carry_dist = dev_zeros<float>(EN);
    st.episode_steps = dev_zeros<int32_t>(E);
    st.reset_count = dev_zeros<int32_t>(E);
    st.prey_loc = dev_zeros<float>((size_t)E * P * 2);
    st.prey_sensed = dev_zeros<uint8_t>((size_t)E * P);
    st.prey_captured = dev_zeros<uint8_t>((size_t)E * P);
    st.loaded = dev_zeros<uint8_t>(EN);
    st.load = dev_zeros<int32_t>(EN);
    st.zone_load = dev_zeros<int32_t>((size_t)E * 2);
    st.messages = dev_zeros<int32_t>((size_t)E * 4);
    st.grid = dev_zeros<uint8_t>((size_t)E * 96);
    st.goal_col = dev_zeros<int32_t>(E);
    st.pixel_type = dev_zeros<uint8_t>(EN);
    st.reached_goal = dev_zeros<uint8_t>(EN);
    st.ep_return = dev_zeros<float>(E);
    st.done_return_sum = dev_zeros<float>(E);
    st.done_count = dev_zeros<int32_t>(E);
    st.done_steps_sum = dev_zeros<int32_t>(E);
    // scratch of the lane-group kernel: every env's next initial state, drawn ahead of time (optional)
    st.next_init = dev_zeros<float>((size_t)E * rg_next_init_stride(&p));
    st.next_episode = dev_zeros<int32_t>(E);
    HIP_OK(hipMemset(st.next_episode, 0xFF, (size_t)E * sizeof(int32_t)));   // -1 = none drawn yet

    rg_step_io io;
    memset(&io, 0, sizeof(io));
    io.obs = dev_zeros<float>(EN * D);
    io.reward = dev_zeros<float>(EN);
    io.done = dev_zeros<uint8_t>(E);
    io.dist_travelled = dev_zeros<float>(EN);
    io.violation = dev_zeros<uint8_t>(E);
    io.remaining = dev_zeros<int32_t>(E);

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    rg_handle *h = rg_create(&p, E, /*env_offset=*/0, /*device=*/0, stream);
    if (!h) {
        fprintf(stderr, "rg_create: %s\n", rg_last_error());
        return 3;
    }
    RG_OK(rg_bind_state(h, &st));
    RG_OK(rg_reset(h, nullptr, /*seed=*/0, /*flags=*/0));

    // a random policy: a few batches of actions, generated on the host once and cycled
    const int n_act = p.scenario == RG_SCN_MATERIAL_TRANSPORT ? 20 : 5, n_batches = 16;
    std::vector<int32_t> host(EN * n_batches);
    uint32_t s = 12345u;
    for (auto &a : host) {
        s = s * 1664525u + 1013904223u;
        a = (int32_t)((s >> 16) % n_act);
    }
    int32_t *actions = dev_zeros<int32_t>(EN * n_batches);
    HIP_OK(hipMemcpy(actions, host.data(), host.size() * sizeof(int32_t), hipMemcpyHostToDevice));

    for (int t = 0; t < 20; ++t) RG_OK(rg_step(h, actions + (t % n_batches) * EN, &io, /*auto_reset=*/1, 0));
    HIP_OK(hipStreamSynchronize(stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < steps; ++t) RG_OK(rg_step(h, actions + (t % n_batches) * EN, &io, 1, 0));
    HIP_OK(hipStreamSynchronize(stream));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<int32_t> counts(E), lens(E);
    std::vector<float> rets(E);
    HIP_OK(hipMemcpy(counts.data(), st.done_count, E * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(lens.data(), st.done_steps_sum, E * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(rets.data(), st.done_return_sum, E * sizeof(float), hipMemcpyDeviceToHost));
    long episodes = 0, len = 0;
    double ret = 0.0;
    for (int e = 0; e < E; ++e) {
        episodes += counts[e];
        len += lens[e];
        ret += rets[e];
    }
    printf("C_ABI_DEMO scenario %d envs %d agents %d steps %d : %.1f us per step, %.3f M agent-steps/s, "
           "%ld episodes finished, mean return %.3f, mean length %.1f\n",
           p.scenario, E, N, steps, sec / steps * 1e6, EN * (double)steps / sec / 1e6, episodes,
           episodes ? ret / episodes : 0.0, episodes ? (double)len / episodes : 0.0);
    RG_OK(rg_destroy(h));
    return episodes > 0 ? 0 : 4;
}
