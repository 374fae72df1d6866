#!/bin/bash
# end of round 3: the whole GPU suite, the driver's bench command, the C5 share rehearsals
set -x
O=gpurun_out/r5g
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
rc=$?
tail -4 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
for N in 2 4 8; do
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N timeout -k 10 300 python bench.py --workload C5 --steps 20 --warmup 4 --no-cpu-baseline --no-regimes --no-live-pmc > $O/C5_share$N.json 2> $O/C5_share$N.err || { tail -5 $O/C5_share$N.err; exit 1; }
done
VR_BENCH_SELF_GATHER=1 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-regimes --no-live-pmc > $O/c3_selfgather.json 2> $O/c3_selfgather.err
python - <<'PY'
import json,glob
d=json.load(open('gpurun_out/r5g/bench_default.json'))
print('driver', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_flavour_resolved'], d['parity']['bit_equal'])
for f in sorted(glob.glob('gpurun_out/r5g/C5*.json'))+['gpurun_out/r5g/c3_selfgather.json']:
    d=json.load(open(f)); p=d.get('pipelined_one_frame_per_launch') or {}
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial'].get('kernel_ms_median'), 'fl', d['config'].get('kernel_flavour_resolved'), '2x1', p.get('ms_per_step'), 'ovl', d['overlapped']['ms_per_step'])
PY
