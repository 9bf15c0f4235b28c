"""The thread-per-env kernel at N = 7 (not part of the library) as a standing check of the round-4 finding
(tools/n7_bisect/README.md): ROCm 7.2 can place register-allocator split copies above the exec-mask restore of an `if`'s join
block, and whether a BUILD has that pattern is visible in its ISA (tools/isa_scan.py exec_prologue).  Four builds of the two N = 7
instantiations (the library's own thread-per-env flags, plain -O3, and the flag set that miscompiled PredatorCapturePrey) are
made ahead of time by `__graft_entry__.build()` (tools/n7_bisect/prebuild.py), so this file needs no compiler on the GPU box.
What must hold on any compiler: a build WITHOUT the pattern is bit-identical to the library's lane-group kernel.  A build WITH
the pattern is never launched (27 of 32 such builds computed wrong results, one ended in a GPU memory fault): it counts as the
scan's prediction "do not ship", which tests/test_kernel_resources.py enforces on the shipped library in the CPU tier."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BISECT = os.path.join(ROOT, "tools", "n7_bisect")
sys.path.insert(0, BISECT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("which", ["mt", "pcp"])
def test_builds_without_the_exec_prologue_pattern_are_bit_identical(which, monkeypatch):
    import prebuild
    ran, not_launched = [], []
    for name, scn, flags in prebuild.VARIANTS:
        if scn != which:
            continue
        lib = prebuild.lib_path(name)
        if not os.path.exists(lib) or not prebuild.up_to_date():
            # no prebuilt probe (a checkout that never ran build()): compile here; a box with neither is a broken setup, not a skip
            from marbler_amd import build as hip_build
            lib = prebuild.compile_variant(name, scn, flags, hipcc=hip_build.hipcc_path())
        findings = prebuild.scan(lib)   # the scan runs on THIS box's copy of the binary, not on the manifest's word
        if findings:
            not_launched.append((name, findings[0][:120]))
            continue
        monkeypatch.setenv("RG_STEP_KERNEL", "group")   # (run_probe sets it itself: restored after the test)
        import run_probe
        res = run_probe.run(lib, which, 192, 3)
        assert res.get("ok"), (name, flags, res)
        ran.append(name)
    print(f"{which}: bit-identical on the GPU: {ran}; flagged by the scan and not launched: {not_launched}")
    assert ran, f"every {which} build has the exec-prologue pattern on this compiler: {not_launched}"
