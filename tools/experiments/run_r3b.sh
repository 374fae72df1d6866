#!/bin/bash
# round 3: A/B of the persistent-wavefront kernel (12) against flavour 6 after the buffer-load change, then the whole GPU suite
set -x
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "persistent or every_variant_every_layout" > $O/pytest_pw.txt 2>&1
rc=$?
tail -3 $O/pytest_pw.txt
[ $rc -eq 0 ] || exit $rc
for fl in 6 12; do
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --flavour $fl --no-cpu-baseline --no-regimes > $O/bench_c3_f$fl.json 2> $O/bench_c3_f$fl.err || exit 1
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --flavour $fl --air noisy --no-cpu-baseline --no-regimes > $O/bench_c3noisy_f$fl.json 2> $O/bench_c3noisy_f$fl.err || exit 1
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --flavour $fl --tf thin --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3thin_f$fl.json 2> $O/bench_c3thin_f$fl.err || exit 1
  timeout -k 10 300 python bench.py --workload C2 --steps 50 --warmup 10 --flavour $fl --no-cpu-baseline --no-live-pmc > $O/bench_c2_f$fl.json 2> $O/bench_c2_f$fl.err || exit 1
  timeout -k 10 300 python bench.py --workload C4 --steps 30 --warmup 5 --flavour $fl --no-cpu-baseline --no-live-pmc > $O/bench_c4_f$fl.json 2> $O/bench_c4_f$fl.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b/bench_*.json')):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    r=d.get('roofline',{})
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'],
          'traffic', r.get('traffic'), 'valu', (r.get('valu') or {}).get('insts_per_launch'), 'ta', (r.get('l1') or {}).get('ta_busy_frac'))
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
echo "rc $?" >> $O/pytest_all.txt
tail -5 $O/pytest_all.txt
