#!/bin/bash
O=gpurun_out/r4g
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_p2_gpu.py tests/test_parity_gpu.py -x -q -m gpu > $O/pytest_p2.txt 2>&1; rc=$?; tail -5 $O/pytest_p2.txt; [ $rc -ne 0 ] && exit 1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0"
for cfg in "C3 17" "C4 17" "C5 17" "C5 6"; do
  set -- $cfg
  timeout -k 10 400 $B --workload $1 --flavour $2 > $O/bench_$1_f$2.json 2> $O/bench_$1_f$2.err || { echo "bench $1 $2 failed"; tail -5 $O/bench_$1_f$2.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4g/bench_*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row, 'fl', d['config']['kernel_flavour_resolved'])
PY
