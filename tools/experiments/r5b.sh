#!/bin/bash
# round 4: the several-frames form of march_p2_kernel for one frame per launch below 4 GiB too?  A/B (-DVR_P2_ALL_BATCH=1)
O=gpurun_out/r5b
mkdir -p $O
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0"
run() {  # tag
  timeout -k 10 300 $B --flavour 17 > $O/c3_f17_$1.json 2> $O/c3_f17_$1.err || exit 1
  timeout -k 10 300 $B --flavour 17 --workload C4 > $O/c4_f17_$1.json 2> $O/c4_f17_$1.err || exit 1
  timeout -k 10 300 $B --flavour 17 --tf thin > $O/thin_f17_$1.json 2> $O/thin_f17_$1.err || exit 1
  timeout -k 10 300 $B --flavour 16 --air noisy > $O/noisy_f16_$1.json 2> $O/noisy_f16_$1.err || exit 1
  timeout -k 10 300 $B --flavour 17 --workload C2 > $O/c2_f17_$1.json 2> $O/c2_f17_$1.err || exit 1
}
run base
VR_EXTRA_HIPCC_FLAGS="-DVR_P2_ALL_BATCH=1" python -c "from volumerendering_amd import build as b; b.build_hip()" > $O/build_b.txt 2>&1 || { tail -5 $O/build_b.txt; exit 1; }
timeout -k 10 300 python -m pytest tests/test_p2_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
run allbatch
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5b/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row, 'fl', d['config']['kernel_flavour_resolved'])
PY
echo done
