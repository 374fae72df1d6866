#!/bin/bash
# flavour 17 after the clean-up: parity (whole files), then against the default on C1 / C2 / C3 / thin
set -x
O=gpurun_out/r4g
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_parity_gpu.py tests/test_random_gpu.py -x -q > $O/pytest.txt 2>&1
rc=$?
tail -4 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
for wl in C1 C2 C3; do
 for fl in 17 0; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --workload $wl --flavour $fl --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_${wl}_f$fl.json 2> $O/bench_${wl}_f$fl.err || { tail -5 $O/bench_${wl}_f$fl.err; exit 1; }
 done
done
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 17 --tf thin --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_thin_f17.json 2> $O/bench_thin_f17.err
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 0 --air noisy --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_noisy_f0.json 2> $O/bench_noisy_f0.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4g/bench_*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'fl', d['config']['kernel_flavour_resolved'], 'arith_ab', d.get('arith_ab',{}).get('serial'))
PY
