#!/bin/bash
# round 3: two steps ahead WITH skipping by whole wavefronts (flavour 17): parity, then against the default on C3 / thin / noisy
set -x
O=gpurun_out/r3s
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_random_gpu.py -x -q -k "persistent or every_variant_every_layout or hostile or fused_every or exact_leaping or random or empty_space" > $O/pytest.txt 2>&1
rc=$?
tail -6 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
for fl in 17 0; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour $fl --no-cpu-baseline --no-regimes > $O/bench_c3_f$fl.json 2> $O/bench_c3_f$fl.err || { tail -5 $O/bench_c3_f$fl.err; exit 1; }
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour $fl --tf thin --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_thin_f$fl.json 2> $O/bench_thin_f$fl.err || { tail -5 $O/bench_thin_f$fl.err; exit 1; }
done
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 17 --identical-frames --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3ident_f17.json 2> $O/bench_c3ident_f17.err || { tail -5 $O/bench_c3ident_f17.err; exit 1; }
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 17 --air noisy --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_noisy_f17.json 2> $O/bench_noisy_f17.err || { tail -5 $O/bench_noisy_f17.err; exit 1; }
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 17 --arith fused --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_fused_f17.json 2> $O/bench_c3_fused_f17.err || { tail -5 $O/bench_c3_fused_f17.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3s/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'fl', d['config']['kernel_flavour_resolved'], 'GB', round((r.get('traffic') or 0)/1e9,2), 'ta', (r.get('l1') or {}).get('ta_busy_frac'), 'valu', (r.get('valu') or {}).get('busy_frac'))
PY
