#!/bin/bash
# the default's lanes-per-ray rule re-tuned (two lanes from 2 rays per lane on): the share rehearsals, C3 and C5
set -x
O=gpurun_out/r4z
mkdir -p $O
for N in 2 4 8; do
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-regimes --no-live-pmc > $O/c3_share$N.json 2> $O/c3_share$N.err || { tail -5 $O/c3_share$N.err; exit 1; }
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N timeout -k 10 300 python bench.py --workload C5 --steps 20 --warmup 4 --no-cpu-baseline --no-regimes --no-live-pmc > $O/C5_share$N.json 2> $O/C5_share$N.err || { tail -5 $O/C5_share$N.err; exit 1; }
done
timeout -k 10 300 python bench.py --workload C1 --steps 30 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc > $O/C1.json 2> $O/C1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4z/*.json')):
    d=json.load(open(f)); p=d.get('pipelined_one_frame_per_launch') or {}
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial'].get('kernel_ms_median'), 'fl', d['config'].get('kernel_flavour_resolved'), '2x1', p.get('ms_per_step'), 'ovl', d['overlapped']['ms_per_step'])
PY
