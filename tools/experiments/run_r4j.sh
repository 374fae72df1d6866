#!/bin/bash
# flavour 17: which packets an XCD class works on (VR_EXP_PW_XCD) and stealing between classes
set -x
O=gpurun_out/r4r
mkdir -p $O
for cfg in "x0s1:VR_EXP_PW_XCD=0 VR_EXP_PW_STEAL=1" "x1s1:VR_EXP_PW_XCD=1 VR_EXP_PW_STEAL=1" "x1s0:VR_EXP_PW_XCD=1 VR_EXP_PW_STEAL=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 17 --no-cpu-baseline --no-regimes > $O/bench_c3_f17_$name.json 2> $O/bench_c3_f17_$name.err || { tail -5 $O/bench_c3_f17_$name.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4r/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'GB', round((r.get('traffic') or 0)/1e9,2), 'l2', r.get('l2_hit_rate'), 'ta', (r.get('l1') or {}).get('ta_busy_frac'))
PY
VR_EXP_PW_XCD=0 VR_EXP_PW_STEAL=1 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "persistent" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
VR_EXP_PW_XCD=0 VR_EXP_PW_STEAL=1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 16 --air noisy --no-cpu-baseline --no-regimes > $O/bench_noisy_f16_x0s1.json 2> $O/bench_noisy_f16_x0s1.err
VR_EXP_PW_XCD=0 VR_EXP_PW_STEAL=1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --tf thin --no-cpu-baseline --no-regimes > $O/bench_thin_x0s1.json 2> $O/bench_thin_x0s1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4r/bench_[nt]*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'fl', d['config']['kernel_flavour_resolved'], 'GB', round((r.get('traffic') or 0)/1e9,2), 'l2', r.get('l2_hit_rate'))
PY
