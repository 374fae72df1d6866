#!/bin/bash
# round 3: how much of a step is memory latency?  The C3 frame and stepping over volumes that fit L2 / the Infinity Cache
set -x
O=gpurun_out/r3g
mkdir -p $O
for n in 32 64 128 256; do for fl in 6 13; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour $fl --vol-n $n --identical-frames --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_n${n}_f$fl.json 2> $O/bench_c3_n${n}_f$fl.err || { tail -5 $O/bench_c3_n${n}_f$fl.err; exit 1; }
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3g/bench_*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'], 'fetched', d['config']['fetched_samples_frame0'], 'comp', d['config']['composited_samples_frame0'])
PY
