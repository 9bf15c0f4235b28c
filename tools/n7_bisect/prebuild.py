"""The four probe builds of tests/test_gpu_n7_probe.py, made ahead of time (by `__graft_entry__.build()`, in the build container)
so that the GPU tier does not depend on a compiler being on the GPU box: tools/n7_bisect/prebuilt/<name>.so + manifest.json
(flags, static findings of tools/isa_scan.py exec_prologue, a hash of the sources they were built from).

    python tools/n7_bisect/prebuild.py [--force]
"""
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "marbler_amd", "csrc")
OUT = os.path.join(HERE, "prebuilt")
MANIFEST = os.path.join(OUT, "manifest.json")
# (name, scenario, flags): the library's own thread-per-env flags, plain -O3, and the flag set under which ROCm 7.2 miscompiled
# the PredatorCapturePrey instantiation (NOTEBOOK.md round 4 #1)
VARIANTS = [("mt_noslp", "mt", "-O3 -fno-slp-vectorize"), ("mt_o3", "mt", "-O3"), ("pcp_noslp", "pcp", "-O3 -fno-slp-vectorize"),
            ("pcp_noslp_nsa", "pcp", "-O3 -fno-slp-vectorize -fno-strict-aliasing -fwrapv -fno-delete-null-pointer-checks")]
SOURCES = [os.path.join(HERE, "tpe_probe.hip")] + [os.path.join(CSRC, h) for h in
                                                   ("step_tpe.h", "device_common.h", "kernel_args.h", "sim_math.h", "ipm_qp.h")] + \
          [os.path.join(ROOT, "include", "robogym.h")]


def source_hash():
    h = hashlib.sha256()
    for p in SOURCES:
        if os.path.exists(p):
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def lib_path(name):
    return os.path.join(OUT, name + ".so")


def compile_variant(name, which, flags, hipcc="/opt/rocm/bin/hipcc"):
    extra = ["-DPROBE_SCN=RG_SCN_PREDATOR_CAPTURE_PREY"] if which == "pcp" else []
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I", CSRC, "-shared",
                           os.path.join(HERE, "tpe_probe.hip"), "-o", lib_path(name)] + flags.split() + extra)
    return lib_path(name)


def scan(path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_scan
    return [m for r in isa_scan.scan_library(path).values() for m in r["exec_prologue"]]


def read_manifest():
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def up_to_date():
    m = read_manifest()
    return m.get("source_hash") == source_hash() and all(os.path.exists(lib_path(n)) for n, _, _ in VARIANTS)


def prebuild(force=False):
    if not force and up_to_date():
        return read_manifest()
    with ThreadPoolExecutor(4) as pool:
        list(pool.map(lambda v: compile_variant(*v), VARIANTS))
    m = {"source_hash": source_hash(), "variants": {}}
    for name, which, flags in VARIANTS:
        found = scan(lib_path(name))
        m["variants"][name] = {"scenario": which, "flags": flags, "exec_prologue_findings": len(found), "first": found[0][:200] if found else ""}
    with open(MANIFEST, "w") as f:
        json.dump(m, f, indent=1, sort_keys=True)
    return m


if __name__ == "__main__":
    print(json.dumps(prebuild(force="--force" in sys.argv), indent=1, sort_keys=True))
