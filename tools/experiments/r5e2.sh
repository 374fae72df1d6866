#!/bin/bash
# (the hand-back and its knobs VR_P2_BACK / VR_EXP_P2_BACK were removed after this measurement: DESIGN 7)
# round 4: the hand-back threshold of the >= 4 GiB kernels (VR_P2_BACK): 8 against 32 on C5 and a rank's half / eighth of it
O=gpurun_out/r5e2
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_p2_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
B="python bench.py --workload C5 --steps 20 --warmup 4 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0 --flavour 17"
run() {  # tag
  timeout -k 10 400 $B > $O/c5_f17_$1.json 2> $O/c5_f17_$1.err || exit 1
  for N in 2 8; do
    VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N timeout -k 10 400 $B > $O/c5_share${N}_f17_$1.json 2> $O/c5_share${N}_f17_$1.err || exit 1
  done
}
run back8
VR_EXTRA_HIPCC_FLAGS="-DVR_P2_BACK=32" python -c "from volumerendering_amd import build as b; b.build_hip()" > $O/build_32.txt 2>&1 || { tail -5 $O/build_32.txt; exit 1; }
run back32
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5e2/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row, 'fl', d['config']['kernel_flavour_resolved'])
PY
echo done
