#!/bin/bash
# round 4: >= 4 GiB volumes through the several-frames form of march_p2_kernel also for one frame per launch: A/B (C5, its shares)
O=gpurun_out/r5a
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_p2_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
B="python bench.py --workload C5 --steps 20 --warmup 4 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0"
run() {  # tag
  timeout -k 10 400 $B --flavour 17 > $O/c5_f17_$1.json 2> $O/c5_f17_$1.err || exit 1
  for N in 2 8; do
    VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N timeout -k 10 400 $B --flavour 17 > $O/c5_share${N}_f17_$1.json 2> $O/c5_share${N}_f17_$1.err || exit 1
  done
}
run new
VR_EXTRA_HIPCC_FLAGS="-DVR_P2_WIN_BATCH=0" python -c "from volumerendering_amd import build as b; b.build_hip()" > $O/build_old.txt 2>&1 || { tail -5 $O/build_old.txt; exit 1; }
run old
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5a/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row, 'fl', d['config']['kernel_flavour_resolved'])
PY
echo done
