#!/bin/bash
# round 3: the bricked HBM layout (layout 0) against round 2's x-fastest arrays (layout 3)
set -x
O=gpurun_out/r3j
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
rc=$?
tail -5 $O/pytest_all.txt
[ $rc -eq 0 ] || exit $rc
for lay in 0 3; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --layout $lay --no-cpu-baseline --no-regimes > $O/bench_c3_l$lay.json 2> $O/bench_c3_l$lay.err || { tail -5 $O/bench_c3_l$lay.err; exit 1; }
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --layout $lay --air noisy --no-cpu-baseline --no-regimes > $O/bench_noisy_l$lay.json 2> $O/bench_noisy_l$lay.err || { tail -5 $O/bench_noisy_l$lay.err; exit 1; }
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --layout $lay --tf thin --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_thin_l$lay.json 2> $O/bench_thin_l$lay.err || { tail -5 $O/bench_thin_l$lay.err; exit 1; }
  for wk in C2 C4 C5; do
    timeout -k 10 400 python bench.py --workload $wk --steps 20 --warmup 5 --layout $lay --no-cpu-baseline --no-live-pmc > $O/bench_${wk}_l$lay.json 2> $O/bench_${wk}_l$lay.err || { tail -5 $O/bench_${wk}_l$lay.err; exit 1; }
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3j/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'value', d['value'], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'], 'fl', d['config']['kernel_flavour_resolved'], 'GB', round((r.get('traffic') or 0)/1e9,2), 'ta', (r.get('l1') or {}).get('ta_busy_frac'), 'l1acc', (r.get('turntable_vs_identical') or {}).get('turntable',{}).get('l1_accesses'))
PY
