#!/bin/bash
# round 3: persistent kernel -- stealing across classes, whole tiles per class (XCD) against packets dealt over the classes
set -x
O=gpurun_out/r3h
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "persistent or mixed" > $O/pytest_pw.txt 2>&1
rc=$?
tail -3 $O/pytest_pw.txt
[ $rc -eq 0 ] || exit $rc
VR_EXP_PW_XCD=0 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "persistent_wavefronts_queue" > $O/pytest_pw_xcd0.txt 2>&1 || { tail -5 $O/pytest_pw_xcd0.txt; exit 1; }
run() { # name, env..., then bench args
  name=$1; shift
  env "$@" > /dev/null 2>&1 || true
}
for cfg in "steal1_xcd1:VR_EXP_PW_STEAL=1:VR_EXP_PW_XCD=1" "steal0_xcd1:VR_EXP_PW_STEAL=0:VR_EXP_PW_XCD=1" "steal1_xcd0:VR_EXP_PW_STEAL=1:VR_EXP_PW_XCD=0" "steal0_xcd0:VR_EXP_PW_STEAL=0:VR_EXP_PW_XCD=0"; do
  name=${cfg%%:*}; rest=${cfg#*:}; e1=${rest%%:*}; e2=${rest#*:}
  env $e1 $e2 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 13 --air noisy --identical-frames --no-cpu-baseline --no-regimes > $O/bench_noisy_$name.json 2> $O/bench_noisy_$name.err || { tail -5 $O/bench_noisy_$name.err; exit 1; }
  env $e1 $e2 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --flavour 13 --identical-frames --no-cpu-baseline --no-regimes > $O/bench_c3_$name.json 2> $O/bench_c3_$name.err || { tail -5 $O/bench_c3_$name.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3h/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'traffic GB', round((r.get('traffic') or 0)/1e9,2), 'l2hit', r.get('l2_hit_rate'), 'ta', (r.get('l1') or {}).get('ta_busy_frac'))
PY
