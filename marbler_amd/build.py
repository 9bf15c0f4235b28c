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
SOURCES = ["robogym_kernels.hip", "robogym_rollout_group.hip", "robogym_kernels_ipm.hip", "robogym_rollout_group_ipm.hip",
           "robogym_tpe.hip", "robogym_rollout_tpe.hip", "robogym_capi.hip", "actor_mfma.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("sim_math.h", "kernel_args.h", "device_common.h", "step_group.h", "step_tpe.h", "step_tpe_ipm.h", "ipm_qp.h", os.path.join("probes", "diag.h"), os.path.join("probes", "actor_diag.h"))
                                           if os.path.exists(os.path.join(CSRC, h))] + \
          [os.path.join(HERE, "..", "include", "robogym.h")]
ARCH = "gfx950"
# Per-file flags (results are bit-identical either way: the same IEEE operations; tools/ab_job.sh measured them).
# * thread-per-env kernels: WITHOUT the SLP vectoriser.  Left on, it pairs the x / y halves of the float arithmetic into
#   v_pk_{add,mul,fma}_f32, which need even-aligned register pairs (more spills at N = 6, a kernel already compiled onto a
#   register budget): N = 6 272 -> 213 us (-22 %) at 524 288 envs, MaterialTransport N = 6 473 -> 386, N = 5 158.5 -> 152.1, N = 4 99 -> 94.
#   What a packed instruction is worth by itself was measured in round 5 (tools/ubench/issue.py, profiles/r5_issue_ubench.txt:
#   independent streams, flop / clk / SIMD): at this kernel's TWO waves per SIMD v_pk_fma_f32 issues 1.43 x the work of two
#   v_fma_f32 (55.8 against 38.9) but v_pk_mul + v_pk_add only 1.06 x the unfused pair (28.2 against 26.5) -- and the step
#   is unfused arithmetic for the most part (-ffp-contract=off; fma only where the float spec says so): the 6-43 % on the
#   instructions that could be paired does not pay for the registers the pairing costs here.  (At ONE wave per SIMD -- the
#   lane-group kernel's regime -- the same instructions are worth 1.86-2.0 x: hence the opposite flag below.)
# * lane-group kernels: the opposite, -mllvm -slp-threshold=-60 (pack wherever possible; 1 083 -> 1 227 packed f32 instructions
#   per translation unit): one wave per SIMD issues an instruction every ~5.5 cycles whatever its width (packed / scalar = 2.02
#   for fma, 1.86 for mul + add in the round-5 microbenchmark), so every pair packed is an issue slot saved -- headline launch
#   13.16 -> 12.73 us, Warehouse 4096 x 8 12.59 -> 12.40, rg_rollout 8.23 -> 7.97 us per step.
# Round 3 found that ONE thread-per-env instantiation (MaterialTransport, N = 7) computes wrong poses under
# -O3 -fno-slp-vectorize.  Round 4 found why (tools/n7_bisect/): ROCm 7.2's register allocator placed
# VGPR -> AGPR live-range split copies at the top of an `if`'s join block BEFORE the `s_or_b64 exec` that restores the exec
# mask -- an SGPR copy of the earlier SGPR allocation sat in front of it and ended what LLVM takes for the block prologue --
# so the saves ran for the `then` lanes only.  A compiler defect, not a property of this source (the same source is clean
# under ASan / UBSan / MSan on the host, tests/test_sanitizers.py); whether a build has it is visible in the ISA, and
# tests/test_kernel_resources.py scans every kernel of the shipped library for it (tools/isa_scan.py exec_prologue).
# The N = 7, 8 instantiations were never dispatched and are no longer built.
GROUP_SLP = ["-mllvm", "-slp-threshold=-60"]
# The interior-point mode's lane-group kernels (their own translation units) are long chains of binary64 arithmetic with
# independent strands beside them: the max-ILP scheduling strategy interleaves the strands (round 5, whole-library A/B on one box:
# interior-point mode 4096 x 5 141.8 -> 134.3 us, Warehouse 4096 x 8 607 -> 548, MaterialTransport 2048 x 6 unchanged; exact mode
# 12.70 -> 12.74, 32 768 envs 27.1 -> 29.0: NOT for the other files).  The one-lane-per-env interior-point kernels stay with the
# default strategy: under max-ILP the N = 5 one shows the exec-prologue shape isa_scan looks for.
IPM_SCHED = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
FILE_FLAGS = {"robogym_tpe.hip": ["-fno-slp-vectorize"], "robogym_rollout_tpe.hip": ["-fno-slp-vectorize"],
              "robogym_kernels.hip": GROUP_SLP, "robogym_rollout_group.hip": GROUP_SLP,
              "robogym_kernels_ipm.hip": GROUP_SLP + IPM_SCHED, "robogym_rollout_group_ipm.hip": GROUP_SLP + IPM_SCHED}
BASE_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
STAMP = LIB + ".flags"


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm 7.x expected under /opt/rocm)")


def flags_stamp(defines=(), extra=(), file_flags=True):
    """What a library was built with, as one string: the rebuild check compares it (a library left behind by an A/B build
    with other flags is stale, whatever its mtime)."""
    per_file = sorted(FILE_FLAGS.items()) if file_flags else []
    return repr((BASE_FLAGS, sorted(defines), list(extra), per_file, SOURCES))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    try:
        with open(STAMP) as f:
            return f.read() != flags_stamp()
    except OSError:
        return True


def build(force=False, verbose=False, defines=(), out=None):
    """defines / out: diagnostic variants (e.g. defines=("RG_STAMPS",), out=".../librobogym_stamps.so").  The environment
    variables RG_EXTRA_HIPCC_FLAGS / RG_NO_FILE_FLAGS (A/B builds, tools/ab_job.sh) apply to such variants only: the shipped
    library is always built with the flags above."""
    extra = os.environ.get("RG_EXTRA_HIPCC_FLAGS", "").split()
    no_file_flags = bool(os.environ.get("RG_NO_FILE_FLAGS"))
    if out is None:
        if not force and not needs_build():
            return LIB
        if extra or no_file_flags:
            # a shell that still exports an A/B flag must not change (or block) the shipped build: its flags are fixed above
            import warnings
            warnings.warn("RG_EXTRA_HIPCC_FLAGS / RG_NO_FILE_FLAGS are for diagnostic variants (build(out=...)) and are IGNORED for the "
                          "shipped library, whose flags are fixed in marbler_amd/build.py")
            extra, no_file_flags = [], False
    shipped = out is None
    out = out or LIB
    # one object per translation unit, compiled side by side (the kernel instantiations dominate:
    # ~2 min each for the lane-group files, ~1 min for the thread-per-env ones), then one link
    objdir = os.path.join(HERE, "build", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    flags = BASE_FLAGS + [f"-D{d}" for d in defines] + extra
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        per_file = [] if no_file_flags else FILE_FLAGS.get(src, [])   # (A/B builds)
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
    with open(out + ".flags", "w") as f:
        f.write(flags_stamp(defines, extra, not no_file_flags) if not shipped else flags_stamp())
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
