#!/bin/bash
# counters of the new march_p2_kernel on C3 (instruction mix, wait states)
O=gpurun_out/r4d
mkdir -p $O
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --pmc-extra --flavour 17 --turn-frames 0 --settle 0 > $O/c3_f17_pmc.json 2> $O/c3_f17_pmc.err || { tail -5 $O/c3_f17_pmc.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4d/c3_f17_pmc.json'))
print('serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'])
for k,v in sorted(d['pmc'].items()):
    print(k, v)
r=d['roofline']; print('traffic', r['traffic'], 'frac', r['frac'], 'l2', r.get('l2_hit_rate'), 'valu', r.get('valu',{}).get('busy_frac'), 'ta', r.get('l1',{}).get('ta_busy_frac'))
PY
