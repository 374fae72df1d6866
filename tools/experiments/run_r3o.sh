#!/bin/bash
# round 3: kernel-trace summary of the one-frame-at-a-time legs only (every leg of the bench one frame at a time)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --no-regimes --steps 40 --warmup 5 --in-flight 1 > $OUT/trace_serial_bench.json 2> $OUT/trace_serial.err || { echo "serial trace failed"; tail -5 $OUT/trace_serial.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$OUT/trace_serial_bench.json").read().strip().splitlines()[-1])
print('serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_hip_events_median'])
PY
grep "march_kernel" $OUT/trace_serial/*/*kernel_stats.csv | head -5
