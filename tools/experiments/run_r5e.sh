#!/bin/bash
# the depth-parallel kernels with the box test behind a wave-uniform guard: parity of the dp flavours, then the shares
set -x
O=gpurun_out/r5e
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_random_gpu.py tests/test_scale_gpu.py -x -q > $O/pytest.txt 2>&1
rc=$?
tail -3 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
for N in 2 4 8; do
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-regimes --no-live-pmc > $O/c3_share$N.json 2> $O/c3_share$N.err || { tail -5 $O/c3_share$N.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5e/*.json')):
    d=json.load(open(f)); p=d.get('pipelined_one_frame_per_launch') or {}
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial'].get('kernel_ms_median'), 'fl', d['config'].get('kernel_flavour_resolved'), '2x1', p.get('ms_per_step'), 'ovl', d['overlapped']['ms_per_step'])
PY
