#!/bin/bash
# flavour 16 (nothing to skip): wavefronts per CU with the round-4 kernel
O=gpurun_out/r4m
mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0 --air noisy --flavour 16"
for t in 384 512 640 768; do
  VR_EXP_P2_THREADS=$t timeout -k 10 300 $B > $O/noisy_t$t.json 2> $O/noisy_t$t.err || { tail -3 $O/noisy_t$t.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4m/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row)
PY
