#!/bin/bash
# (the hand-back and its knobs VR_P2_BACK / VR_EXP_P2_BACK were removed after this measurement: DESIGN 7)
# round 4: the pipelined loop hands a run of >= VR_P2_BACK identity steps back to the approach loop: A/B against -DVR_P2_BACK=0
O=gpurun_out/r5d
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_p2_gpu.py tests/test_configs_gpu.py tests/test_parity_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0"
run() {  # tag
  timeout -k 10 300 $B --flavour 17 > $O/c3_f17_$1.json 2> $O/c3_f17_$1.err || exit 1
  timeout -k 10 300 $B --flavour 17 --workload C4 > $O/c4_f17_$1.json 2> $O/c4_f17_$1.err || exit 1
  timeout -k 10 300 $B --flavour 17 --tf thin > $O/thin_f17_$1.json 2> $O/thin_f17_$1.err || exit 1
  timeout -k 10 300 $B --flavour 17 --workload C2 > $O/c2_f17_$1.json 2> $O/c2_f17_$1.err || exit 1
  timeout -k 10 400 $B --flavour 17 --workload C5 --steps 20 > $O/c5_f17_$1.json 2> $O/c5_f17_$1.err || exit 1
}
run back8
VR_EXTRA_HIPCC_FLAGS="-DVR_P2_BACK=0" python -c "from volumerendering_amd import build as b; b.build_hip()" > $O/build_old.txt 2>&1 || { tail -5 $O/build_old.txt; exit 1; }
run back0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5d/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row, 'fl', d['config']['kernel_flavour_resolved'])
PY
echo done
