"""Build librobogym_hip.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m marbler_amd.build

hipcc cross-compiles without a GPU.  -ffp-contract=off: the kernels' float arithmetic is an
explicit sequence of IEEE operations (csrc/sim_math.h) that the CPU oracle reproduces bit for
bit; implicit fma contraction would break that.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librobogym_hip.so")
SOURCES = ["robogym_kernels.hip", "robogym_tpe.hip", "robogym_capi.hip"]
HEADERS = [os.path.join(CSRC, "sim_math.h"), os.path.join(CSRC, "kernel_args.h"), os.path.join(CSRC, "device_common.h"),
           os.path.join(HERE, "..", "include", "robogym.h")]
ARCH = "gfx950"


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm 7.x expected under /opt/rocm)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, defines=(), out=None):
    """defines / out: diagnostic variants (e.g. defines=("RG_STAMPS",), out=".../librobogym_stamps.so")."""
    if out is None and not force and not needs_build():
        return LIB
    out = out or LIB
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-fPIC", "-shared", "-std=c++17",
           "-ffp-contract=off", "-Wall", "-Wno-unused-function"] + [f"-D{d}" for d in defines] + ["-o", out] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
