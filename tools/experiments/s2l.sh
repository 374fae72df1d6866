#!/bin/bash
# A trip's two steps shaded as one block in the lit kernel without skipping (flavour 16; shade_blend_pair): the GPU suite, then noisy air
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2l
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --turn-frames 0"
$B --air noisy > $O/c3_noisy.json 2> $O/c3_noisy.err; echo "c3 noisy rc $?"
$B --air noisy --arith fused --no-live-pmc > $O/c3_noisy_fused.json 2> $O/c3_noisy_fused.err; echo "c3 noisy fused rc $?"
$B --tf thin --flavour 16 --no-live-pmc > $O/c3_thin_f16.json 2> $O/c3_thin_f16.err; echo "c3 thin f16 rc $?"
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s2l/*.json')):
    t=open(f).read(); i=t.find('{"metric')
    if i<0: print(f,'no line'); continue
    d=json.loads(t[i:].splitlines()[0])
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'pipelined', d['pipelined_one_frame_per_launch']['ms_per_step'], 'batched', d['overlapped']['ms_per_step'], 'ran', d['serial']['kernel_choice'], 'traffic', d['roofline'].get('traffic'), d['roofline'].get('frac'))
P
