#!/bin/bash
# round 4: (1) how much of a C3 frame of march_p2_kernel is memory latency (volumes that fit L2 / the Infinity Cache),
# (2) launch shapes of the persistent kernel with launches in flight and with several frames per launch
O=gpurun_out/r4c
mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0"
for vn in 64 128 256; do
  timeout -k 10 300 $B --flavour 17 --vol-n $vn > $O/voln_${vn}_f17.json 2> $O/voln_${vn}_f17.err || { tail -5 $O/voln_${vn}_f17.err; exit 1; }
done
timeout -k 10 300 $B --flavour 6 --vol-n 64 > $O/voln_64_f6.json 2> $O/voln_64_f6.err || exit 1
i=0
for cfg in "768 1 0" "768 1 1" "384 2 1" "384 2 0" "512 1 1" "256 2 1" "384 1 1"; do
  set -- $cfg
  VR_EXP_P2_THREADS=$1 VR_EXP_P2_WGS=$2 VR_EXP_P2_DYNQ=$3 timeout -k 10 300 $B --flavour 17 > $O/shape_$1_$2_$3.json 2> $O/shape_$1_$2_$3.err || { tail -5 $O/shape_$1_$2_$3.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c/*.json')):
    d=json.load(open(f))
    row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row, 'fl', d['config']['kernel_flavour_resolved'], 'fetched', d['config']['fetched_samples_frame0'])
PY
echo done
