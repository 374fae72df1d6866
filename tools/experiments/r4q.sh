#!/bin/bash
# flavour 18 (march_kernel + slot tables in LDS) and the indexed loads of every kernel: parity, then every leg on C3 / C5 / C2
O=gpurun_out/r4q
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_p2_gpu.py tests/test_parity_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; tail -6 $O/pytest.txt; [ $rc -ne 0 ] && exit 1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0"
for cfg in "C3 6 exact0" "C3 18 exact0" "C5 6 exact0" "C5 18 exact0" "C2 6 exact0" "C2 18 exact0" "C3 18 noisy"; do
  set -- $cfg
  timeout -k 10 400 $B --workload $1 --flavour $2 --air $3 > $O/bench_$1_f$2_$3.json 2> $O/bench_$1_f$2_$3.err || { echo "bench $cfg failed"; tail -5 $O/bench_$1_f$2_$3.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4q/bench_*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row, d['config']['kernel_flavour_resolved'])
PY
