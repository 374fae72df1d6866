#!/bin/bash
# round 4: workgroups of 4 wavefronts, three per CU (the same 3 wavefronts per SIMD as one workgroup of 12, but a CU's
# share frees up a third at a time for the next launch in flight)
O=gpurun_out/r4t
mkdir -p $O
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0"
for cfg in "768 1" "256 3" "768 1" "256 3"; do
  set -- $cfg
  i=$((i+1))
  VR_EXP_P2_THREADS=$1 VR_EXP_P2_WGS=$2 timeout -k 10 300 $B --flavour 17 > $O/shape_$1_$2_$i.json 2> $O/shape_$1_$2_$i.err || { tail -5 $O/shape_$1_$2_$i.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4t/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row, 'fl', d['config']['kernel_flavour_resolved'])
PY
echo done
