#!/bin/bash
# round 3: flavour 17 with the per-step vote: parity, C3 / thin with counters
set -x
O=gpurun_out/r4d
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_random_gpu.py -x -q -k "persistent or every_variant_every_layout or hostile or fused_every or exact_leaping or random or empty_space" > $O/pytest.txt 2>&1
rc=$?
tail -6 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
for tf in default thin; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 17 --tf $tf --no-cpu-baseline --no-regimes > $O/bench_${tf}_f17.json 2> $O/bench_${tf}_f17.err || { tail -5 $O/bench_${tf}_f17.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4d/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'fl', d['config']['kernel_flavour_resolved'], 'GB', round((r.get('traffic') or 0)/1e9,2), 'ta', (r.get('l1') or {}).get('ta_busy_frac'), 'valu', (r.get('valu') or {}).get('busy_frac'), 'insts', (r.get('valu') or {}).get('insts_per_launch'), 'ovl', d['overlapped']['ms_per_step'])
PY
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 16 --air noisy --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_noisy_f16.json 2> $O/bench_noisy_f16.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4d/bench_noisy_f16.json')); print('noisy f16', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'])
PY
timeout -k 10 300 python tools/block_trace.py --flavour 17 > $O/trace_c3_f17.txt 2>&1; tail -9 $O/trace_c3_f17.txt
