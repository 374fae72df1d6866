#!/bin/bash
O=gpurun_out/r4s
mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 --settle 0 --flavour 17"
for p in 0 1; do
  VR_EXP_P2_PRIO=$p timeout -k 10 300 $B > $O/c3_prio$p.json 2> $O/c3_prio$p.err || { tail -3 $O/c3_prio$p.err; exit 1; }
  VR_EXP_P2_PRIO=$p timeout -k 10 300 $B --workload C4 > $O/c4_prio$p.json 2> $O/c4_prio$p.err || { tail -3 $O/c4_prio$p.err; exit 1; }
  VR_EXP_P2_PRIO=$p timeout -k 10 300 $B --tf thin > $O/thin_prio$p.json 2> $O/thin_prio$p.err || { tail -3 $O/thin_prio$p.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4s/*.json')):
    d=json.load(open(f)); row=[f.split('/')[-1]]
    for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
        s=d.get(k)
        if s: row += [k[:6], s['ms_per_step'], s['kernel_ms_median']]
    print(*row)
PY
