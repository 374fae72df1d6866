#!/bin/bash
set -x
O=gpurun_out/r3n
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "lds_tiles_by or hostile" > $O/pytest_lt.txt 2>&1
rc=$?
tail -5 $O/pytest_lt.txt
[ $rc -eq 0 ] || exit $rc
for fl in 15; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --flavour $fl --no-cpu-baseline --no-regimes > $O/bench_c3_f$fl.json 2> $O/bench_c3_f$fl.err || { tail -5 $O/bench_c3_f$fl.err; exit 1; }
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour $fl --air noisy --no-cpu-baseline --no-regimes > $O/bench_noisy_f$fl.json 2> $O/bench_noisy_f$fl.err || { tail -5 $O/bench_noisy_f$fl.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3n/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'value', d['value'], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'GB', round((r.get('traffic') or 0)/1e9,2), 'ta', (r.get('l1') or {}).get('ta_busy_frac'), 'valu', (r.get('valu') or {}).get('insts_per_launch'))
PY
