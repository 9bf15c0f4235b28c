"""CPU sanitizer tier: AddressSanitizer + UndefinedBehaviorSanitizer builds of the two
pieces of native code that run on the host -- the C oracle (oracle/oracle.c, `make -C oracle asan`) and the C ABI's
host half (csrc/robogym_capi.hip compiled --offload-host-only) -- driven by the existing CPU test files in a child
python with the sanitizer runtime preloaded.  Any ASan / UBSan report aborts the child (-fno-sanitize-recover).
GPU sanitizers are not available on this pool.  Round 4: the thread-per-env step kernel -- per-lane scalar C++ apart from its
LDS staging copy and the fused reset -- is therefore compiled for the HOST from the shipped device headers (tests/sanitize/:
64 threads stand in for the 64 lanes) and run for every scenario and N = 2..8 against the float32 oracle, bit for bit, under
ASan + UBSan and under MSan.  The lane-group kernel's arithmetic is the same float spec but lives in DPP lane permutes and
has no host form; it is covered by the bit-exact parity tests and the static ISA checks of tests/test_kernel_resources.py."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPTS = "detect_leaks=0:abort_on_error=1:halt_on_error=1:verify_asan_link_order=0"


def _pytest_child(args, env):
    e = dict(os.environ, ASAN_OPTIONS=OPTS, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    return r.stdout


def test_oracle_under_asan_ubsan():
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    lib = os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so")
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    out = _pytest_child(["tests/test_oracle_golden.py::test_c_oracle_f64_matches_reference_vectors",
                         "tests/test_oracle_spec.py"], {"ORACLE_LIB": lib, "LD_PRELOAD": rt})
    assert " passed" in out


def test_c_abi_host_half_under_asan_ubsan():
    from marbler_amd import build as hip_build
    try:
        hip_build.hipcc_path()
    except RuntimeError:
        pytest.skip("no hipcc")
    lib = hip_build.build_host_sanitized()
    out = _pytest_child(["tests/test_host.py", "-k", "exports_every_declared_symbol or rejects_bad_parameters or no_cpu_fallback"],
                        {"ROBOGYM_LIB": lib, "LD_PRELOAD": hip_build.asan_runtime()})
    assert " passed" in out


@pytest.mark.parametrize("mode", ["asan", "msan"])
def test_thread_per_env_kernel_on_the_host_under_sanitizers(mode):
    """csrc/step_tpe.h (+ device_common.h, sim_math.h, kernel_args.h as shipped) compiled as host C++ against
    tests/sanitize/hip_shim, every scenario x N = 2..8 (29 instantiations, plus the multi-step form for N = 5, 7 and the gymma
    block), ragged batches, free-running with auto-reset: every output and state word equal to the float32 oracle's and no
    report from AddressSanitizer + UndefinedBehaviorSanitizer (out-of-bounds and misaligned accesses, shifts, signed
    overflow, float-to-int casts out of range) or MemorySanitizer (a branch, address or output that depends on an
    uninitialised value -- the class of error that would put 'a float's bits into an int counter').  This is how
    round 4 established that the N = 7 miscompute of round 3 is not in this source."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "sanitize"))
    import host_sim
    try:
        host_sim.clang()
    except RuntimeError as exc:
        pytest.skip(str(exc))
    n, r = host_sim.run(mode)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert f"{n} cases, 0 mismatches" in r.stdout, tail
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, tail


def test_host_simulation_sees_an_uninitialised_read_and_an_overrun():
    """The tier's own smoke alarm: the same harness with one state array left uninitialised (MSan build) or one output
    array a row short (ASan build) must fail."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "sanitize"))
    import host_sim
    try:
        host_sim.clang()
    except RuntimeError as exc:
        pytest.skip(str(exc))
    for mode, fault, needle in (("msan", "uninit", "MemorySanitizer"), ("asan", "overrun", "AddressSanitizer")):
        n, r = host_sim.run(mode, fault=fault, only_first=True)
        assert r.returncode != 0 and needle in r.stderr, (mode, r.stdout[-500:], r.stderr[-1500:])
