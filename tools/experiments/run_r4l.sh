#!/bin/bash
# lane-mask ballots everywhere: the whole GPU suite, then the main regimes
set -x
O=gpurun_out/r4l
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
rc=$?
tail -3 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-regimes --no-live-pmc "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; exit 1; }; }
run c3_default
run c3_f6 --flavour 6
run thin --tf thin
run noisy --air noisy
run C2 --workload C2
run C4 --workload C4
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4l/bench_*.json')):
    d=json.load(open(f)); p=d.get('pipelined_one_frame_per_launch') or {}
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'fl', d['config']['kernel_flavour_resolved'], '2x1', p.get('ms_per_step'), '2x4', d['overlapped']['ms_per_step'])
PY
