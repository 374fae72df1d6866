#!/bin/bash
# round 3: lanes per ray per packet (flavour 14): parity, then the serial leg of C3 by threshold
set -x
O=gpurun_out/r3f
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "mixed or every_variant_every_layout or hostile or fused_every or persistent" > $O/pytest_mixed.txt 2>&1
rc=$?
tail -15 $O/pytest_mixed.txt
[ $rc -eq 0 ] || exit $rc
for pct in 60 75 90; do
  VR_EXP_SPLIT_PCT=$pct timeout -k 10 300 python bench.py --steps 40 --warmup 10 --flavour 14 --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_f14_p$pct.json 2> $O/bench_c3_f14_p$pct.err || { tail -5 $O/bench_c3_f14_p$pct.err; exit 1; }
done
timeout -k 10 300 python bench.py --steps 40 --warmup 10 --flavour 6 --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_f6.json 2> $O/bench_c3_f6.err || exit 1
VR_EXP_SPLIT_PCT=75 timeout -k 10 300 python bench.py --steps 40 --warmup 10 --flavour 14 --identical-frames --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_f14_same.json 2> $O/bench_c3_f14_same.err || exit 1
timeout -k 10 300 python bench.py --steps 40 --warmup 10 --flavour 6 --identical-frames --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_f6_same.json 2> $O/bench_c3_f6_same.err || exit 1
for wk in C2 C4; do for fl in 6 14; do
  timeout -k 10 300 python bench.py --workload $wk --steps 40 --warmup 10 --flavour $fl --no-cpu-baseline --no-live-pmc > $O/bench_${wk}_f$fl.json 2> $O/bench_${wk}_f$fl.err || exit 1
done; done
timeout -k 10 300 python tools/block_trace.py --flavour 14 > $O/trace_c3_f14.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3f/bench_*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['value'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'])
PY
