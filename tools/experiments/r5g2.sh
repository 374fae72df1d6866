#!/bin/bash
# round 4 (after the one code form): threads per workgroup of flavour 16 on noisy air once more
O=gpurun_out/r5g2
mkdir -p $O
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0 --flavour 16 --air noisy"
for t in 512 768 640 512 768; do
  i=$((i+1))
  VR_EXP_P2_THREADS=$t timeout -k 10 300 $B > $O/noisy_t${t}_$i.json 2> $O/noisy_t${t}_$i.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5g2/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row)
PY
