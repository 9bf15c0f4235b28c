"""The thread-per-env kernel at N = 7 (no longer part of the library) as a standing check of the round-4 finding
(tools/n7_bisect/README.md): the instantiations are built HERE, with the flags the library's thread-per-env files use and with the
flag set under which ROCm 7.2 miscompiled one of them, scanned statically (tools/isa_scan.py exec_prologue) and run against the
library's lane-group kernel.  What must hold on any compiler: a build WITHOUT the pattern is bit-identical.  A build WITH the
pattern is not launched at all (it may compute with stale registers: 27 of 32 such builds gave wrong results, one a memory fault)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BISECT = os.path.join(ROOT, "tools", "n7_bisect")


@pytest.mark.parametrize("which,flags", [("mt", "-O3 -fno-slp-vectorize"), ("mt", "-O3"), ("pcp", "-O3 -fno-slp-vectorize"),
                                         ("pcp", "-O3 -fno-slp-vectorize -fno-strict-aliasing -fwrapv -fno-delete-null-pointer-checks")])
def test_a_build_without_the_exec_prologue_pattern_is_bit_identical(which, flags, tmp_path, monkeypatch):
    from marbler_amd import build as hip_build
    try:
        hipcc = hip_build.hipcc_path()
    except RuntimeError:
        pytest.skip("no hipcc on this box")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, BISECT)
    import isa_scan
    lib = str(tmp_path / f"{which}_probe.so")
    extra = ["-DPROBE_SCN=RG_SCN_PREDATOR_CAPTURE_PREY"] if which == "pcp" else []
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "marbler_amd", "csrc"),
                           "-shared", os.path.join(BISECT, "tpe_probe.hip"), "-o", lib] + flags.split() + extra)
    findings = [m for r in isa_scan.scan_library(lib).values() for m in r["exec_prologue"]]
    if findings:
        # NEVER launched: a build with the pattern computes with stale registers -- wrong poses at best, wild addresses at worst
        # (round 4: this very build of PredatorCapturePrey N = 7 ended in a GPU memory fault)
        pytest.skip(f"this compiler's build has the exec-prologue pattern and is not run: {findings[0][:160]}")
    monkeypatch.setenv("RG_STEP_KERNEL", "group")   # (run_probe sets it itself: restored after the test)
    import run_probe
    res = run_probe.run(lib, which, 192, 3)
    assert res.get("ok"), (which, flags, res)
