"""The C ABI used without Python: examples/c_abi_demo.cpp (a plain HIP host program) compiles and links
against include/robogym.h + librobogym_hip.so here (no GPU needed), and on a GPU box steps 2048 envs
for 300 steps through rg_create / rg_bind_state / rg_reset / rg_step and reports finished episodes."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "c_abi_demo")
PARAMS = os.path.join(ROOT, "examples", "pcp_params.bin")


def _build():
    from marbler_amd import build as hip_build
    from marbler_amd.params import load_config, make_params, params_to_bytes
    src = os.path.join(ROOT, "examples", "c_abi_demo.cpp")
    lib = os.path.join(ROOT, "marbler_amd", "librobogym_hip.so")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        subprocess.check_call([hip_build.hipcc_path(), "--offload-arch=gfx950", "-O2", "-I" + os.path.join(ROOT, "include"), src,
                               "-L" + os.path.join(ROOT, "marbler_amd"), "-lrobogym_hip",
                               "-Wl,-rpath,$ORIGIN/../marbler_amd", "-o", EXE])
    cfg = load_config("PredatorCapturePrey", overrides={"predator": 3, "capture": 2, "n_agents": 5})
    with open(PARAMS, "wb") as f:
        f.write(params_to_bytes(make_params("PredatorCapturePrey", cfg)))


def test_c_demo_compiles_and_links_against_the_header():
    _build()
    assert os.path.exists(EXE)
    # without arguments it prints its usage and exits 1 -- before any HIP call
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr


@pytest.mark.gpu
def test_c_demo_steps_envs_from_plain_c():
    import torch
    if torch.cuda.is_initialized():   # never start a program from a process that has initialised the GPU
        pytest.skip("this process already uses the GPU; run tests/test_c_abi_demo.py on its own")
    _build()
    r = subprocess.run([EXE, PARAMS, "2048", "300"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    m = re.search(r"C_ABI_DEMO scenario 0 envs 2048 agents 5 steps 300 : ([0-9.]+) us per step, ([0-9.]+) M agent-steps/s, "
                  r"(\d+) episodes finished, mean return (-?[0-9.]+), mean length ([0-9.]+)", r.stdout)
    assert m, r.stdout
    assert int(m.group(3)) > 2048 and 20.0 < float(m.group(5)) <= 81.0 and float(m.group(4)) < 0.0
