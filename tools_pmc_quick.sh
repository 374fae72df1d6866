#!/bin/bash
# usage: pmc_quick.sh <tag> <bench args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > /dev/null 2> $OUT/pmc_$N.err || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.err; }
done
python3 $R/tools_pmc_summary.py $OUT /tmp/pq_$TAG "x" C3 default
