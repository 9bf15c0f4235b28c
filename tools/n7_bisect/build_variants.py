"""Builds variants of tools/n7_bisect/tpe_probe.hip side by side: name=flags pairs on the command line, or --bisect LO HI COUNT
(COUNT -opt-bisect-limit values spread over [LO, HI]) on top of the failing flag set.

    python tools/n7_bisect/build_variants.py good="-O3" bad="-O3 -fno-slp-vectorize"
    python tools/n7_bisect/build_variants.py --bisect 0 226254 30
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "build")
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "marbler_amd", "csrc"), "-shared",
        os.path.join(HERE, "tpe_probe.hip")]
FAIL = "-O3 -fno-slp-vectorize"


def build(item):
    name, flags = item
    out = os.path.join(OUT, name + ".so")
    r = subprocess.run(BASE + flags.split() + ["-o", out], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(f"{name}: {r.stderr[-800:]}\n")
    return name, r.returncode


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    items = []
    args = sys.argv[1:]
    prefix = "mt"
    extra = ""
    keep = "--keep" in args
    if keep:
        args.remove("--keep")
    if "--pcp" in args:
        args.remove("--pcp")
        prefix, extra = "pcp", " -DPROBE_SCN=RG_SCN_PREDATOR_CAPTURE_PREY"
    if "--base" in args:
        i = args.index("--base")
        FAIL = args[i + 1]
        del args[i:i + 2]
    if args and args[0] == "--bisect":
        lo, hi, cnt = int(args[1]), int(args[2]), int(args[3])
        ks = sorted({lo + (hi - lo) * i // (cnt - 1) for i in range(cnt)})
        items = [(f"{prefix}_bisect_{k:07d}", f"{FAIL}{extra} -mllvm -opt-bisect-limit={k}") for k in ks]
    else:
        items = [(f"{prefix}_{a.split('=', 1)[0]}", a.split("=", 1)[1] + extra) for a in args]
    for f in os.listdir(OUT):
        if not keep and f.endswith(".so"):
            os.remove(os.path.join(OUT, f))
    with ThreadPoolExecutor(8) as pool:
        for name, rc in pool.map(build, items):
            print(name, "built" if rc == 0 else "FAILED")
