#!/bin/bash
# round 3: LDS tiles by LDS-DMA (flavour 15): parity, then against flavour 6
set -x
O=gpurun_out/r3m
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "lds_tiles_by or every_variant_every_layout or hostile or exact_leaping or fused_every or fused_empty" > $O/pytest_lt.txt 2>&1
rc=$?
tail -15 $O/pytest_lt.txt
[ $rc -eq 0 ] || exit $rc
for fl in 15 6; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --flavour $fl --no-cpu-baseline --no-regimes > $O/bench_c3_f$fl.json 2> $O/bench_c3_f$fl.err || { tail -5 $O/bench_c3_f$fl.err; exit 1; }
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour $fl --air noisy --no-cpu-baseline --no-regimes > $O/bench_noisy_f$fl.json 2> $O/bench_noisy_f$fl.err || { tail -5 $O/bench_noisy_f$fl.err; exit 1; }
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour $fl --tf thin --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_thin_f$fl.json 2> $O/bench_thin_f$fl.err || { tail -5 $O/bench_thin_f$fl.err; exit 1; }
done
timeout -k 10 300 python tools/block_trace.py --flavour 15 > $O/trace_c3_f15.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3m/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'value', d['value'], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'], 'GB', round((r.get('traffic') or 0)/1e9,2), 'ta', (r.get('l1') or {}).get('ta_busy_frac'), 'valu', (r.get('valu') or {}).get('insts_per_launch'))
PY
