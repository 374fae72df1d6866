#!/bin/bash
# round 4: the approach loop in the depth-parallel kernels (vr_dp.h): parity, then a rank's share of C3 / C5 with them forced
O=gpurun_out/r5c
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
run() {  # tag
  for N in 4 8; do
    for fl in 10 11; do
      VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0 --flavour $fl > $O/c3_share${N}_f${fl}_$1.json 2> $O/c3_share${N}_f${fl}_$1.err || exit 1
    done
  done
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=8 timeout -k 10 300 python bench.py --workload C5 --steps 20 --warmup 4 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0 --flavour 11 > $O/c5_share8_f11_$1.json 2> $O/c5_share8_f11_$1.err || exit 1
}
run new
VR_EXTRA_HIPCC_FLAGS="-DVR_APPROACH=0" python -c "from volumerendering_amd import build as b; b.build_hip()" > $O/build_old.txt 2>&1 || { tail -5 $O/build_old.txt; exit 1; }
run old
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row, 'fl', d['config']['kernel_flavour_resolved'])
PY
echo done
