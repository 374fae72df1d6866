#!/bin/bash
# round 3: where does the one-frame-at-a-time leg stand per kernel form (C3), and what do the timelines look like
set -x
O=gpurun_out/r3c
mkdir -p $O
for fl in 8 11 10; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour $fl --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_f$fl.json 2> $O/bench_c3_f$fl.err || exit 1
done
VR_EXP_PW_LTF=0 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 12 --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_f12_noltf.json 2> $O/bench_c3_f12_noltf.err || exit 1
for fl in 6 12 11; do
  timeout -k 10 300 python tools/block_trace.py --flavour $fl > $O/trace_c3_f$fl.txt 2>&1 || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c/bench_*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'])
PY
