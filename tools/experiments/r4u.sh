#!/bin/bash
# round 4: rays parked from the pipelined loop (vr_p2.h) -- parity tests of the p2 kernels, then C3 / C4 / noisy / C5 timings
O=gpurun_out/r4u
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_p2_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0"
timeout -k 10 300 $B --flavour 17 > $O/c3_f17.json 2> $O/c3_f17.err || exit 1
timeout -k 10 300 $B --flavour 17 --workload C4 > $O/c4_f17.json 2> $O/c4_f17.err || exit 1
timeout -k 10 300 $B --flavour 16 --air noisy > $O/noisy_f16.json 2> $O/noisy_f16.err || exit 1
timeout -k 10 300 $B --flavour 17 --tf thin > $O/thin_f17.json 2> $O/thin_f17.err || exit 1
timeout -k 10 300 $B --flavour 17 --workload C5 --steps 20 > $O/c5_f17.json 2> $O/c5_f17.err || exit 1
timeout -k 10 300 $B --flavour 17 --workload C2 > $O/c2_f17.json 2> $O/c2_f17.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4u/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row, 'fl', d['config']['kernel_flavour_resolved'])
PY
echo done
