#!/bin/bash
# Where the cycles of the step kernels go: SQ wait / issue / instruction-fetch counters, one rocprofv3 --pmc pass per
# group (counters only: no trace domains in the same run).
#   gpurun --timeout 900 -- 'bash tools/stall_job.sh <tag>'
set -o pipefail
TAG=${1:-stall}
SET=${2:-sq}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline"
pass() {
  local name=$1; shift
  timeout -k 10 100 rocprofv3 --pmc "$@" -d $OUT/$name -o p -- $BENCH > $OUT/$name.log 2>&1
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name timed out"; exit 9; fi
  [ $rc -ne 0 ] && { echo "pass $name rc=$rc"; tail -5 $OUT/$name.log; }
  return 0
}
if [ "$SET" = sq ]; then
  NAMES="wait issue ifetch mix level"
  pass wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
  pass issue SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
  pass ifetch SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
  pass mix SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
  pass level SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_INST_REQ
else  # the vector-memory pipeline: SQ issue -> TA -> TCP (L1) -> TCC (L2)
  # (a pass that asks one block for more counters than it has aborts inside rocprofv3 and then sits until the timeout:
  # two per TA pass, four per TCP / TCC pass)
  NAMES="ta1 ta2 tcp1 tcp2 tcc1"
  pass ta1 TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE
  pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
  pass tcp1 TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
  pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum
  pass tcc1 TCC_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
fi
ARGS=""
for n in $NAMES; do
  db=$(find $OUT/$n -name "*.db" | head -1)
  [ -n "$db" ] && ARGS="$ARGS --pmc $n=$db"
done
python tools/summarize_rocpd.py $TAG $ARGS && cp profiles/${TAG}_pmc_summary.csv $OUT/
find $OUT -name "*.db" -size +8M -delete
echo JOB_DONE
