#!/bin/bash
# (the hand-back and its knobs VR_P2_BACK / VR_EXP_P2_BACK were removed after this measurement: DESIGN 7)
# round 4: C5, the hand-back to the approach loop on / off, alternating twice on one box (VR_EXP_P2_BACK)
O=gpurun_out/r5f2
mkdir -p $O
B="python bench.py --workload C5 --steps 20 --warmup 4 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0 --flavour 17"
for r in 1 2; do
  for b in 8 0 32; do
    VR_EXP_P2_BACK=$b timeout -k 10 400 $B > $O/c5_back${b}_$r.json 2> $O/c5_back${b}_$r.err || exit 1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5f2/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row)
PY
echo done
