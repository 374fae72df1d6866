#!/bin/bash
# which packets an XCD works on: per packet (1, the default), per 16x16 sub-block (2), per tile (0)
set -x
O=gpurun_out/r5i
mkdir -p $O
for x in 1 2 0; do
  VR_EXP_XCD=$x timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-regimes > $O/bench_c3_x$x.json 2> $O/bench_c3_x$x.err || { tail -5 $O/bench_c3_x$x.err; exit 1; }
  VR_EXP_XCD=$x timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 6 --no-cpu-baseline --no-regimes > $O/bench_c3_f6_x$x.json 2> $O/bench_c3_f6_x$x.err || { tail -5 $O/bench_c3_f6_x$x.err; exit 1; }
done
VR_EXP_XCD=2 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --air noisy --no-cpu-baseline --no-regimes > $O/bench_noisy_x2.json 2> $O/bench_noisy_x2.err
VR_EXP_XCD=2 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --tf thin --no-cpu-baseline --no-regimes > $O/bench_thin_x2.json 2> $O/bench_thin_x2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5i/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']; p=d.get('pipelined_one_frame_per_launch') or {}
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'fl', d['config']['kernel_flavour_resolved'], 'GB', round((r.get('traffic') or 0)/1e9,2), 'l2', r.get('l2_hit_rate'), '2x1', p.get('ms_per_step'), 'ovl', d['overlapped']['ms_per_step'])
PY
