#!/bin/bash
# Flavour 15 (LDS tiles by LDS-DMA) as part of the shipped library: the whole GPU suite, then what it costs on C3 (default table and
# noisy air) beside the default's choice -- one frame at a time, 2 launches in flight, batched
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2f
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --turn-frames 0"
$B --flavour 15 > $O/c3_f15.json 2> $O/c3_f15.err; echo "c3 f15 rc $?"
$B --flavour 15 --air noisy > $O/c3_noisy_f15.json 2> $O/c3_noisy_f15.err; echo "c3 noisy f15 rc $?"
$B --air noisy > $O/c3_noisy.json 2> $O/c3_noisy.err; echo "c3 noisy rc $?"
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s2f/*.json')):
    t=open(f).read(); i=t.find('{"metric')
    if i<0: print(f,'no line'); continue
    d=json.loads(t[i:].splitlines()[0])
    print(f.split('/')[-1], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'pipelined', d['pipelined_one_frame_per_launch']['ms_per_step'], 'batched', d['overlapped']['ms_per_step'], 'ran', d['serial']['kernel_choice'], 'traffic', d['roofline'].get('traffic'), 'parity', d['parity']['bit_equal'])
P
