#!/bin/bash
# Round-5 measurement call: GPU test tier, the final bench line, kernel traces + PMC passes of the headline and the other
# BASELINE configurations, the driver's own command line, the actor kernel's trace + counters, the evaluation loop.
#   gpurun --timeout 1100 -- 'bash tools/r5_measure_job.sh <tag> [tests] [headline] [configs] [driver] [actor] [eval]'
set -o pipefail
TAG=${1:-r5m}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT $OUT/summ; export TMPDIR=/tmp
ALL="$*"; [[ -z "$ALL" ]] && ALL="tests headline configs ipm driver actor eval runner"
has() { [[ " $ALL " == *" $1 "* ]]; }

if has tests; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest -m gpu rc=$?: $(tail -1 $OUT/gputest.log)"
fi

prof() {  # name, bench args ...: bench line, kernel trace, three PMC passes of the same command
  local name=$1; shift
  local B="python3 bench.py --no-cpu-baseline --no-saturated $*"
  timeout -k 10 200 $B > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 3; }
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt -- $B > $OUT/kt_$name.log 2>&1 || { tail -5 $OUT/kt_$name.log; exit 3; }
  find $OUT/kt_$name -name "*kernel_trace.csv" -size +20M -delete
  local P="python3 bench.py --no-cpu-baseline --no-saturated --steps 300 --warmup 50 $*"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pf_$name -o p -- $P > $OUT/pf_$name.log 2>&1 || { tail -5 $OUT/pf_$name.log; exit 4; }
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pw_$name -o p -- $P > $OUT/pw_$name.log 2>&1 || { tail -5 $OUT/pw_$name.log; exit 4; }
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/ps_$name -o p -- $P > $OUT/ps_$name.log 2>&1 || { tail -5 $OUT/ps_$name.log; exit 4; }
  python3 tools/summarize_rocpd.py r5_$name --outdir $OUT/summ --stats $(find $OUT/kt_$name -name "*.db") \
      --pmc fetch=$(find $OUT/pf_$name -name "*.db") --pmc write=$(find $OUT/pw_$name -name "*.db") --pmc sq=$(find $OUT/ps_$name -name "*.db") > /dev/null || exit 4
  rm -rf $OUT/kt_$name $OUT/pf_$name $OUT/pw_$name $OUT/ps_$name
  echo "$name: $(python3 -c "import json;d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]);print('%.4g agent-steps/s, %.2f us/step' % (d['value'], d['ms_per_step']*1e3))")"
}

if has headline; then
  # the full default line (cpu baseline, saturated legs, actor leg) once, then the profiled command (no side legs in the trace)
  timeout -k 10 500 python3 bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err || { tail -5 $OUT/final_bench.err; exit 2; }
  python3 -c "
import json
d = json.loads(open('$OUT/final_bench.json').read().strip().splitlines()[-1])
print('bench: %.4g agent-steps/s, %.2f us/step; roofline frac %.4f; saturated %s; actor %s; ipm %s' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['frac'],
      {k: round(v, 4) for k, v in d['saturated'].items() if k in ('hbm_frac', 'ms_per_step')}, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['actor']['roofline'].items() if k in ('achieved', 'frac')}, round(d['interior_point_mode']['ms_per_step'] * 1e3, 1)))
"
  prof final --no-graph
fi
if has configs; then
  prof mt_2048x6 --scenario MaterialTransport --envs-per-gpu 2048 --steps 1000
  prof warehouse_4096x8 --scenario Warehouse --steps 1000
  prof mt_4096x6 --scenario MaterialTransport --steps 1000
  prof pcp_32768x5 --envs-per-gpu 32768 --steps 1000
fi
if has ipm; then   # the headline workload in the interior-point mode (barrier_solver: cvxopt), and the BASELINE shapes in both modes
  prof ipm_pcp_4096x5 --barrier-solver cvxopt --steps 400 --warmup 40
  timeout -k 10 300 python3 tools/ipm_probe.py > $OUT/ipm_probe.jsonl 2> $OUT/ipm_probe.err || { tail -5 $OUT/ipm_probe.err; exit 7; }
  cat $OUT/ipm_probe.jsonl
  timeout -k 10 200 python3 tests/ipm_bench.py > $OUT/ipm_bench.jsonl 2>&1 || { tail -5 $OUT/ipm_bench.jsonl; exit 7; }
  grep '"N"' $OUT/ipm_bench.jsonl
fi
if has runner; then
  timeout -k 10 200 python3 tools/runner_probe.py 4096 200 > $OUT/runner_probe.txt 2>&1 || { tail -5 $OUT/runner_probe.txt; exit 8; }
  grep hidden $OUT/runner_probe.txt
fi
if has driver; then   # the driver's own command line (20 timed steps after 5 warm-up steps), six times
  for i in 1 2 3 4 5 6; do timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 >> $OUT/driver_like.jsonl 2>> $OUT/driver_like.err || { tail -5 $OUT/driver_like.err; exit 9; }; done
  python3 -c "
import json
for l in open('$OUT/driver_like.jsonl'):
    if l.startswith('{'):
        d = json.loads(l); print('driver-like: %.4g agent-steps/s, %.2f us/step, kernel %.2f us' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['kernel_ms_avg'] * 1e3))
"
fi
if has actor; then
  timeout -k 10 200 python3 tools/actor_probe.py > $OUT/actor_probe.txt 2>&1 || { tail -5 $OUT/actor_probe.txt; exit 5; }
  grep hidden $OUT/actor_probe.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_actor -o kt -- python3 tools/actor_probe.py > $OUT/kt_actor.log 2>&1 || { tail -5 $OUT/kt_actor.log; exit 5; }
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $OUT/ps_actor -o p -- python3 tools/actor_probe.py > $OUT/ps_actor.log 2>&1 || { tail -5 $OUT/ps_actor.log; exit 5; }
  python3 tools/summarize_rocpd.py r5_actor --outdir $OUT/summ --stats $(find $OUT/kt_actor -name "*.db") --pmc sq=$(find $OUT/ps_actor -name "*.db") | grep -i "actor" | head -12
  rm -rf $OUT/kt_actor $OUT/ps_actor
fi
if has eval; then
  timeout -k 10 300 python3 tools/eval_probe.py > $OUT/eval_probe.txt 2>&1 || { tail -5 $OUT/eval_probe.txt; exit 6; }
  cat $OUT/eval_probe.txt | grep -v amdgpu.ids
fi
echo JOB_DONE
