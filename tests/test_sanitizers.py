"""CPU sanitizer tier: AddressSanitizer + UndefinedBehaviorSanitizer builds of the two
pieces of native code that run on the host -- the C oracle (oracle/oracle.c, `make -C oracle asan`) and the C ABI's
host half (csrc/robogym_capi.hip compiled --offload-host-only) -- driven by the existing CPU test files in a child
python with the sanitizer runtime preloaded.  Any ASan / UBSan report aborts the child (-fno-sanitize-recover).
GPU sanitizers are not available on this pool; device code is covered by the bit-exact parity tests instead."""
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
