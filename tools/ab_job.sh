#!/bin/bash
# A/B of diagnostic library builds on one box: tools/perf_probe.py --set <set> once per library, interleaved twice.
#   gpurun --timeout 900 -- 'bash tools/ab_job.sh <tag> <set> <libA.so|shipped> <libB.so|shipped> ...'
set -o pipefail
TAG=$1; SET=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  for lib in "$@"; do
    if [[ "$lib" == shipped ]]; then unset ROBOGYM_LIB; else export ROBOGYM_LIB=$PWD/marbler_amd/$lib; fi
    timeout -k 10 300 python3 tools/perf_probe.py --set $SET >> $OUT/$SET.jsonl 2>> $OUT/$SET.err || { tail -5 $OUT/$SET.err; exit 2; }
  done
done
python3 - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/$SET.jsonl") if l.startswith("{")]
d = collections.OrderedDict()
for r in rows:
    key = (r.get("scenario"), r.get("E"), r.get("api", "rg_step"))
    d.setdefault(key, collections.OrderedDict()).setdefault(r.get("lib", "shipped").split("/")[-1], []).append(r["us_per_step"])
for key, libs in d.items():
    print(key, "  ".join(f"{lib}: {min(v):.2f} us (runs {', '.join('%.2f' % x for x in v)})" for lib, v in libs.items()))
PY
echo JOB_DONE
