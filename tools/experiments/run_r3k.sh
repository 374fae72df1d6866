#!/bin/bash
# round 3: kernel forms on the bricked layout (which one should the default pick where?)
set -x
O=gpurun_out/r3k
mkdir -p $O
for fl in 6 13 12; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --flavour $fl --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_f$fl.json 2> $O/bench_c3_f$fl.err || { tail -5 $O/bench_c3_f$fl.err; exit 1; }
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour $fl --tf thin --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_thin_f$fl.json 2> $O/bench_thin_f$fl.err || { tail -5 $O/bench_thin_f$fl.err; exit 1; }
done
for fl in 6 13; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour $fl --air noisy --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_noisy_f$fl.json 2> $O/bench_noisy_f$fl.err || { tail -5 $O/bench_noisy_f$fl.err; exit 1; }
  timeout -k 10 300 python bench.py --workload C2 --steps 40 --warmup 8 --flavour $fl --no-cpu-baseline --no-live-pmc > $O/bench_C2_f$fl.json 2> $O/bench_C2_f$fl.err || { tail -5 $O/bench_C2_f$fl.err; exit 1; }
  timeout -k 10 400 python bench.py --workload C5 --steps 12 --warmup 4 --flavour $fl --no-cpu-baseline --no-live-pmc > $O/bench_C5_f$fl.json 2> $O/bench_C5_f$fl.err || { tail -5 $O/bench_C5_f$fl.err; exit 1; }
done
for fl in 6 12; do
  timeout -k 10 300 python bench.py --workload C4 --steps 30 --warmup 8 --flavour $fl --no-cpu-baseline --no-live-pmc > $O/bench_C4_f$fl.json 2> $O/bench_C4_f$fl.err || { tail -5 $O/bench_C4_f$fl.err; exit 1; }
done
timeout -k 10 300 python tools/block_trace.py --flavour 6 > $O/trace_c3_f6.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3k/bench_*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], 'value', d['value'], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'])
PY
