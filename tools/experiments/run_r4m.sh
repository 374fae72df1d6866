#!/bin/bash
# the two-steps-ahead kernels with the last steps of a packet in a loop of their own: parity, then the regimes
set -x
O=gpurun_out/r4p
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_parity_gpu.py tests/test_random_gpu.py tests/test_scale_gpu.py -x -q > $O/pytest.txt 2>&1
rc=$?
tail -3 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-regimes --no-live-pmc "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; exit 1; }; }
run c3_default
run thin_f17 --tf thin --flavour 17
run noisy --air noisy
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4p/bench_*.json')):
    d=json.load(open(f)); p=d.get('pipelined_one_frame_per_launch') or {}
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'fl', d['config']['kernel_flavour_resolved'], 'fused', d.get('arith_ab',{}).get('serial'))
PY
