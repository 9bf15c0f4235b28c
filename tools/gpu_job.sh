#!/bin/bash
# One gpurun call of the round's standard evidence: GPU tests, bench line, kernel trace, PMC passes.
#   gpurun --timeout 1100 -- 'bash tools/gpu_job.sh <tag> [notest] [nopmc]'
set -o pipefail
TAG=${1:-job}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [[ " $* " != *" notest "* ]]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
  tail -3 $OUT/pytest.log
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 7; }
  tail -1 $OUT/smoke.log
fi
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 2; }
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms %.5f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for k in ("rollout", "saturated"):
    if k in d: print(k, "%.4g agent-steps/s, %.5f ms" % (d[k]["agent_steps_per_s"], d[k]["ms_per_step"]))
PY
if [[ " $* " != *" nopmc "* ]]; then
  BENCH="python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-graph"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1 || { tail -20 $OUT/kt.log; exit 3; }
  BENCH="python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-graph"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 4; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $BENCH > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 5; }
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- $BENCH > $OUT/pmc_sq.log 2>&1 || { tail -20 $OUT/pmc_sq.log; exit 6; }
  # condensed on the box (the rocpd databases are too large to merge back): $OUT/summ/<tag>_{kernel_stats,pmc_summary}.csv
  python3 tools/summarize_rocpd.py $TAG --outdir $OUT/summ --stats $(find $OUT/kt -name "*.db") --pmc fetch=$(find $OUT/pmc_fetch -name "*.db") \
      --pmc write=$(find $OUT/pmc_write -name "*.db") --pmc sq=$(find $OUT/pmc_sq -name "*.db") > /dev/null || exit 8
  rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
  grep "step_kernel" $OUT/summ/${TAG}_kernel_stats.csv | cut -c1-160
fi
echo JOB_DONE
