#!/bin/bash
# Profiling recipe used for profiles/ (run on the GPU box via gpurun): kernel trace + stats, then PMC passes
# (each counter group in its own run, never combined with other trace domains).
# usage: tools_prof.sh <tag> [bench workload args, e.g. --workload C3 --tf thin]
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" > $OUT/bench_trace.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
echo "trace done"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > /dev/null 2> $OUT/pmc_$N.err || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.err; }
  echo "pmc $N done"
done
echo done
