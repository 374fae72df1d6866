#!/bin/bash
# The waits for the sorts a launch depends on: on the host before the enqueue (VR_EXP_HOST_ORDER_WAIT=1) or on the launch's stream (0)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2o
mkdir -p $O
cd $R
./tools/ubench/stream_gap > $O/stream_gap.txt 2>&1
for H in 0 1; do
  export VR_EXP_HOST_ORDER_WAIT=$H
  python3 tools/launch_gap.py --run 2>&1 | grep frames
  for W in C3 C1 C2; do
    python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 > $O/${W}_$H.json 2> $O/${W}_$H.err
    python3 - <<P
import json
t=open('$O/${W}_$H.json').read(); d=json.loads(t[t.find('{"metric'):].splitlines()[0])
print('$W host-wait $H: serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'present', d['serial_with_present']['ms_per_step'], 'pipelined', d['pipelined_one_frame_per_launch']['ms_per_step'], 'batched', d['overlapped']['ms_per_step'], 'sync', d['sync_8d']['t_frame_ms_median'])
P
  done
done
