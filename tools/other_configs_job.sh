#!/bin/bash
# The other BASELINE configurations on one MI355X: bench lines + one rocprofv3 --kernel-trace --stats run each.
#   gpurun --timeout 600 -- 'bash tools/other_configs_job.sh <tag>'
set -o pipefail
TAG=${1:-other}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-saturated "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 2; }
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt -- python3 bench.py --no-cpu-baseline --no-saturated "$@" > $OUT/kt_$name.log 2>&1 || { tail -5 $OUT/kt_$name.log; exit 3; }
  find $OUT/kt_$name -name "*kernel_trace.csv" -size +20M -delete
}
run warehouse_4096x8 --scenario Warehouse
run mt_4096x6 --scenario MaterialTransport --steps 1000
run mt_2048x6 --scenario MaterialTransport --envs-per-gpu 2048 --steps 1000
run pcp_32768x5 --envs-per-gpu 32768 --steps 1000
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["config"]["workload"] if "workload" in d["config"] else d["config"], "%.4g %s" % (d["value"], d["unit"]), "%.2f us/step" % (d["ms_per_step"] * 1e3), "frac %.4f" % d["roofline"]["frac"])
PY
echo JOB_DONE
