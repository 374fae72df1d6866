#!/bin/bash
# the default (measured choice) on every configuration and leg, and the rank-share rehearsals
O=gpurun_out/r4h
mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0"
for w in C1 C2 C3 C4 C5; do
  timeout -k 10 400 $B --workload $w > $O/bench_$w.json 2> $O/bench_$w.err || { echo "bench $w failed"; tail -5 $O/bench_$w.err; exit 1; }
done
timeout -k 10 300 $B --air noisy > $O/bench_C3noisy.json 2> $O/bench_C3noisy.err || exit 1
timeout -k 10 300 $B --tf thin > $O/bench_C3thin.json 2> $O/bench_C3thin.err || exit 1
for w in C3 C5; do
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=8 timeout -k 10 300 $B --workload $w > $O/share8_$w.json 2> $O/share8_$w.err || { tail -5 $O/share8_$w.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4h/*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1])
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: print('   ', k[:6], s['ms_per_step'], s['kernel_ms_median'], s.get('kernel_choice'))
PY
