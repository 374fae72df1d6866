#!/bin/bash
# a rank's share of the tiles (rehearsal on one GPU): the default against flavour 17 and 12
set -x
O=gpurun_out/r4t
mkdir -p $O
for N in 2 4 8; do
 for fl in 0 17; do
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N timeout -k 10 300 python bench.py --steps 60 --warmup 10 --flavour $fl --no-cpu-baseline --no-regimes --no-live-pmc > $O/c3_share${N}_f$fl.json 2> $O/c3_share${N}_f$fl.err || { tail -5 $O/c3_share${N}_f$fl.err; exit 1; }
 done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4t/*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial'].get('kernel_ms_median'), 'fl', d['config'].get('kernel_flavour_resolved'), 'ovl', d['overlapped']['ms_per_step'], d.get('rank0_stage_timeline'))
PY
