#!/bin/bash
# round 3: whole GPU suite after the policy change, default bench of the regimes the policy touches
set -x
O=gpurun_out/r3i
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
rc=$?
tail -5 $O/pytest_all.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --air noisy --no-cpu-baseline --no-regimes > $O/bench_noisy.json 2> $O/bench_noisy.err || { tail -5 $O/bench_noisy.err; exit 1; }
timeout -k 10 300 python bench.py --workload C4 --steps 30 --warmup 8 --no-cpu-baseline > $O/bench_C4.json 2> $O/bench_C4.err || { tail -5 $O/bench_C4.err; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > $O/bench_c3.json 2> $O/bench_c3.err || { tail -5 $O/bench_c3.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'value', d['value'], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'], 'fl', d['config']['kernel_flavour_resolved'], 'traffic GB', round((r.get('traffic') or 0)/1e9,2))
    if 'regimes' in d: print([ (x['air'],x['tf'],x['kernel_ms'],x['kernel_flavour_resolved']) for x in d['regimes']])
PY
