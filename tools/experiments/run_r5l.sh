#!/bin/bash
# is there a cliff in the default's choice along a long turntable (chains crossing 400 samples -> flavour 12)?
set -x
O=gpurun_out/r5l
mkdir -p $O
for fl in 0 17; do
  timeout -k 10 300 python bench.py --steps 60 --warmup 8 --flavour $fl --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_60_f$fl.json 2> $O/bench_c3_60_f$fl.err || { tail -5 $O/bench_c3_60_f$fl.err; exit 1; }
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --tf thin --flavour $fl --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_thin_40_f$fl.json 2> $O/bench_thin_40_f$fl.err || { tail -5 $O/bench_thin_40_f$fl.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5l/bench_*.json')):
    d=json.load(open(f)); s=d['serial']
    print(f.split('/')[-1], 'serial', s['ms_per_step'], s['kernel_ms_median'], s['kernel_ms_p10_p90'], s.get('kernel_ms_mean'), 'fl(last)', d['config']['kernel_flavour_resolved'])
PY
