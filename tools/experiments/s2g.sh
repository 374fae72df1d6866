#!/bin/bash
# order_blocks_kernel with four records' loads in flight per thread: how long it holds its CU, how late the next launch's last
# workgroup starts, the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/launch_gap.py --run > $O/run.txt 2>&1; grep -v amdgpu.ids $O/run.txt
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/launch_gap.py --run > $O/run_traced.txt 2>&1 || { tail -5 $O/run_traced.txt; exit 1; }
python3 $R/tools/launch_gap.py --read $O/kt > $O/gap.txt 2>&1; cat $O/gap.txt; rm -rf $O/kt
cd $R
timeout -k 10 300 python3 -m pytest tests/test_parity_gpu.py -x -q -k "order or persistent or records" > $O/pytest_order.txt 2>&1; tail -2 $O/pytest_order.txt
for i in 1 2; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 > $O/bench$i.json 2> $O/bench$i.err
  python3 - <<P
import json
t=open('$O/bench$i.json').read(); d=json.loads(t[t.find('{"metric'):].splitlines()[0])
print('bench $i: serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'pipelined', d['pipelined_one_frame_per_launch']['ms_per_step'], 'batched', d['overlapped']['ms_per_step'], 'sync', d['sync_8d']['t_frame_ms_median'])
P
done
