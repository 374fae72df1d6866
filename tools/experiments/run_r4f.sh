#!/bin/bash
# flavours 16 / 17: wavefronts per CU (the in-flight corner data of 12 wavefronts is six times the L1)
set -x
O=gpurun_out/r4f
mkdir -p $O
for t in 768 640 512 384 256; do
  VR_EXP_P2_THREADS=$t timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 17 --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_f17_t$t.json 2> $O/bench_c3_f17_t$t.err || { tail -5 $O/bench_c3_f17_t$t.err; exit 1; }
  VR_EXP_P2_THREADS=$t timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 16 --air noisy --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_noisy_f16_t$t.json 2> $O/bench_noisy_f16_t$t.err || { tail -5 $O/bench_noisy_f16_t$t.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4f/bench_*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'fl', d['config']['kernel_flavour_resolved'])
PY
