#!/bin/bash
# round 3: the pipelined persistent kernel (13) against 6 / 12; counter calibration of the issue-rate microbenchmark
set -x
O=gpurun_out/r3d
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "persistent or every_variant_every_layout or exact_leaping or hostile or fused_every" > $O/pytest_pw.txt 2>&1
rc=$?
tail -3 $O/pytest_pw.txt
[ $rc -eq 0 ] || exit $rc
for fl in 13 12 6; do
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --flavour $fl --no-cpu-baseline --no-regimes > $O/bench_c3_f$fl.json 2> $O/bench_c3_f$fl.err || exit 1
done
for fl in 13; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --flavour $fl --air noisy --no-cpu-baseline --no-regimes > $O/bench_c3noisy_f$fl.json 2> $O/bench_c3noisy_f$fl.err || exit 1
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --flavour $fl --tf thin --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3thin_f$fl.json 2> $O/bench_c3thin_f$fl.err || exit 1
  timeout -k 10 300 python bench.py --workload C2 --steps 50 --warmup 10 --flavour $fl --no-cpu-baseline --no-live-pmc > $O/bench_c2_f$fl.json 2> $O/bench_c2_f$fl.err || exit 1
  timeout -k 10 300 python bench.py --workload C5 --steps 10 --warmup 3 --flavour $fl --no-cpu-baseline --no-live-pmc > $O/bench_c5_f$fl.json 2> $O/bench_c5_f$fl.err || exit 1
done
timeout -k 10 300 python bench.py --workload C5 --steps 10 --warmup 3 --flavour 6 --no-cpu-baseline --no-live-pmc > $O/bench_c5_f6.json 2> $O/bench_c5_f6.err || exit 1
timeout -k 10 300 python tools/block_trace.py --flavour 13 > $O/trace_c3_f13.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d/bench_*.json')):
    d=json.load(open(f))
    r=d.get('roofline',{})
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'],
          'traffic', r.get('traffic'), 'valu', (r.get('valu') or {}).get('insts_per_launch'), 'ta', (r.get('l1') or {}).get('ta_busy_frac'))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/ubench_pmc -- $GRAFT_REPO_ROOT/tools/ubench/valu_issue > $GRAFT_REPO_ROOT/$O/ubench_pmc.txt 2>&1
echo "rocprof rc $?"
