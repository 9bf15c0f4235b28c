#!/bin/bash
# Round-3 measurement call: counter calibration, per-configuration kernel traces + PMC passes, phase stamps, pair-test probe.
#   gpurun --timeout 1100 -- 'bash tools/r3_measure_job.sh <tag> [calib] [configs] [stamps] [probe]'
# Needs (built on the CPU box beforehand, they travel with the snapshot): tools/ubench/libhbm_calib.so,
# marbler_amd/librobogym_stamps.so (-DRG_STAMPS), marbler_amd/librobogym_nopair.so (-DRG_PROBE_NO_PAIRTEST).
set -o pipefail
TAG=${1:-r3m}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
want() { [[ " $* " == *" $1 "* ]]; }
ALL="$*"; [[ -z "$ALL" ]] && ALL="calib configs stamps probe floor"
has() { [[ " $ALL " == *" $1 "* ]]; }

if has calib; then
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/calib_f -o p -- python3 tools/ubench/hbm_calib.py > $OUT/calib_f.log 2>&1 || { tail -20 $OUT/calib_f.log; exit 2; }
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/calib_w -o p -- python3 tools/ubench/hbm_calib.py > $OUT/calib_w.log 2>&1 || { tail -20 $OUT/calib_w.log; exit 2; }
  python3 tools/ubench/hbm_calib.py --tag r3 --outdir $OUT/summ --summarize $(find $OUT/calib_f $OUT/calib_w -name "*.db") || exit 2
  rm -rf $OUT/calib_f $OUT/calib_w
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
  python3 tools/summarize_rocpd.py r3_$name --outdir $OUT/summ --stats $(find $OUT/kt_$name -name "*.db") \
      --pmc fetch=$(find $OUT/pf_$name -name "*.db") --pmc write=$(find $OUT/pw_$name -name "*.db") --pmc sq=$(find $OUT/ps_$name -name "*.db") > /dev/null || exit 4
  rm -rf $OUT/kt_$name $OUT/pf_$name $OUT/pw_$name $OUT/ps_$name
  echo "$name: $(python3 -c "import json;d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]);print('%.4g agent-steps/s, %.2f us/step' % (d['value'], d['ms_per_step']*1e3))")"
}
if has configs; then
  prof mt_2048x6 --scenario MaterialTransport --envs-per-gpu 2048 --steps 1000
  prof warehouse_4096x8 --scenario Warehouse --steps 1000
  prof mt_4096x6 --scenario MaterialTransport --steps 1000
  prof pcp_32768x5 --envs-per-gpu 32768 --steps 1000
fi

if has driver; then   # the driver's own command line (20 timed steps after 5 warm-up steps), three times
  for i in 1 2 3; do timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 >> $OUT/driver_like.jsonl 2>> $OUT/driver_like.err || { tail -5 $OUT/driver_like.err; exit 9; }; done
  python3 -c "
import json
for l in open('$OUT/driver_like.jsonl'):
    if l.startswith('{'):
        d = json.loads(l); print('driver-like: %.4g agent-steps/s, %.2f us/step, kernel %.2f us' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['kernel_ms_avg'] * 1e3))
"
fi

if has stamps; then
  for cfg in "4096 PredatorCapturePrey" "2048 MaterialTransport" "4096 Warehouse"; do
    timeout -k 10 120 python3 tools/stamp_probe.py $cfg >> $OUT/stamps.txt 2>&1 || { tail -5 $OUT/stamps.txt; exit 5; }
  done
  cat $OUT/stamps.txt
fi

if has probe; then
  timeout -k 10 200 python3 tools/perf_probe.py --set pairtest > $OUT/pairtest_shipped.jsonl 2> $OUT/pairtest.err || { tail -5 $OUT/pairtest.err; exit 6; }
  ROBOGYM_LIB=$PWD/marbler_amd/librobogym_nopair.so timeout -k 10 200 python3 tools/perf_probe.py --set pairtest > $OUT/pairtest_nopair.jsonl 2>> $OUT/pairtest.err || { tail -5 $OUT/pairtest.err; exit 6; }
  paste -d'\n' $OUT/pairtest_shipped.jsonl $OUT/pairtest_nopair.jsonl
fi
if has floor; then
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/floor_f -o p -- python3 tools/fetch_floor_probe.py > $OUT/floor_f.log 2>&1 || { tail -5 $OUT/floor_f.log; exit 7; }
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/floor_w -o p -- python3 tools/fetch_floor_probe.py > $OUT/floor_w.log 2>&1 || { tail -5 $OUT/floor_w.log; exit 7; }
  timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/floor_i -o p -- python3 tools/fetch_floor_probe.py > $OUT/floor_i.log 2>&1 || { tail -5 $OUT/floor_i.log; }
  python3 tools/summarize_rocpd.py r3_fetch_floor --outdir $OUT/summ --pmc fetch=$(find $OUT/floor_f -name "*.db") --pmc write=$(find $OUT/floor_w -name "*.db") \
      $(f=$(find $OUT/floor_i -name "*.db"); [[ -n "$f" ]] && echo "--pmc cache=$f") | grep step_kernel
  rm -rf $OUT/floor_f $OUT/floor_w $OUT/floor_i
fi
echo JOB_DONE
