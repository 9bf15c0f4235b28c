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
SOURCES = ["robogym_kernels.hip", "robogym_rollout_group.hip", "robogym_tpe.hip", "robogym_rollout_tpe.hip",
           "robogym_tpe_hi.hip", "robogym_rollout_tpe_hi.hip", "robogym_capi.hip", "actor_mfma.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("sim_math.h", "kernel_args.h", "device_common.h", "step_group.h", "step_tpe.h")] + \
          [os.path.join(HERE, "..", "include", "robogym.h")]
ARCH = "gfx950"
# Per-file flags.  The thread-per-env kernels are compiled WITHOUT the SLP vectoriser: left on, it pairs the x / y halves of
# the float arithmetic into v_pk_{add,mul,fma}_f32, which need even-aligned register pairs (more spills at N = 6, a kernel
# already compiled onto a register budget) and issue no faster than the two scalar instructions on gfx950 -- measured at
# 524 288 envs: N = 6 272 -> 213 us (-22 %), MaterialTransport N = 6 473 -> 386, N = 5 158.5 -> 152.1, N = 4 99 -> 94.  The
# lane-group kernels keep it: there one wave per SIMD runs a dependent chain, fewer instructions on the chain win, and the
# same flag makes the headline launch 4.6 % SLOWER (13.24 -> 13.85 us).  Results are bit-identical either way (same IEEE
# operations); tools/ab_job.sh / tools/tpe_ab_probe.py with RG_EXTRA_HIPCC_FLAGS=-fno-slp-vectorize is how it was measured.
# The N = 7, 8 instantiations (robogym_*tpe_hi.hip: never dispatched by the library, reachable with RG_STEP_KERNEL=tpe only)
# keep the default flags: built with -O3 -fno-slp-vectorize, ONE of them -- MaterialTransport, N = 7 -- computes wrong poses
# from the first step on (5 of 935 GPU tests; every other instantiation passes; the same source passes with -O3 and the
# vectoriser, with -O2 -fno-slp-vectorize, and fails again when compiled for two waves per SIMD; with -fno-strict-aliasing
# -fwrapv -fno-delete-null-pointer-checks added it passes and PredatorCapturePrey N = 7 fails instead: it is N = 7 that is sensitive; capped at one QP
# sweep the failing kernel returns a float's bit pattern in its integer sweep counter: a register-assignment matter).  Neither a use of
# undefined behaviour in the source nor a hardware hazard was found in the time available; until it is explained those two
# files stay on the flags every test and 250 M fuzzed env steps have covered.
# The lane-group kernels get the opposite treatment: -mllvm -slp-threshold=-60 makes the vectoriser pack wherever it can (1 083 ->
# 1 227 packed f32 instructions per translation unit) instead of where its cost model sees a profit -- on a dependent chain at one
# wave per SIMD every instruction saved is ~5 cycles: headline launch 13.16 -> 12.73 us, Warehouse 4096 x 8 12.59 -> 12.40,
# rg_rollout 8.23 -> 7.97 us per step; the MaterialTransport N = 6 instantiation grows from 128 to 130 VGPRs (four waves per SIMD
# -> three), which costs 3.5 % at 32 768 x 6 envs, a size between the BASELINE shapes (2048 / 4096 x 6: -0.4 %).  Thresholds
# -12 / -24 / -60 / -200 measured within 1 % of each other (tools/ab_job.sh).
GROUP_SLP = ["-mllvm", "-slp-threshold=-60"]
FILE_FLAGS = {"robogym_tpe.hip": ["-fno-slp-vectorize"], "robogym_rollout_tpe.hip": ["-fno-slp-vectorize"],
              "robogym_kernels.hip": GROUP_SLP, "robogym_rollout_group.hip": GROUP_SLP}


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
    # one object per translation unit, compiled side by side (the kernel instantiations dominate:
    # ~2 min each for the lane-group files, ~1 min for the thread-per-env ones), then one link
    objdir = os.path.join(HERE, "build", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall",
             "-Wno-unused-function"] + [f"-D{d}" for d in defines] + list(os.environ.get("RG_EXTRA_HIPCC_FLAGS", "").split())
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        per_file = [] if os.environ.get("RG_NO_FILE_FLAGS") else FILE_FLAGS.get(src, [])   # (A/B builds)
        cmd = [hipcc_path()] + flags + per_file + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return out


def build_host_sanitized(out=None):
    """CPU sanitizer tier: the C ABI's host half (csrc/robogym_capi.hip) compiled for the HOST ONLY with
    ASan + UBSan, linked against tests/sanitize/launch_stubs.cpp in place of the device translation units.
    Used by tests/test_sanitizers.py for the no-GPU cases of tests/test_host.py; never shipped.  (GPU ASan is
    not available on this pool: sanitizers run on the CPU build only.)"""
    out = out or os.path.join(HERE, "build", "librobogym_capi_asan.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    stubs = os.path.join(HERE, "..", "tests", "sanitize", "launch_stubs.cpp")
    cmd = [hipcc_path(), "-x", "hip", "--offload-host-only", "-O1", "-g", "-fno-omit-frame-pointer",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fPIC", "-shared", "-std=c++17",
           "-I", CSRC, os.path.join(CSRC, "robogym_capi.hip"), stubs, "-o", out]
    subprocess.check_call(cmd)
    return out


def asan_runtime():
    """clang's shared ASan runtime (to LD_PRELOAD into the python process that loads the sanitized library)."""
    return subprocess.check_output([hipcc_path(), "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
