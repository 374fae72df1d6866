#!/bin/bash
# After the unlit kernel's 16 wavefronts per CU: the whole GPU suite, then the C1 / C2 profile lines again (tools/profiles_r04.sh's commands)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2i
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for W in C1 C2; do
  python3 bench.py --workload $W --steps 30 --warmup 5 --no-regimes --no-cpu-baseline > $O/${W}_default.json 2> $O/${W}_default.err; echo "$W rc $?"
  python3 bench.py --workload $W --tf thin --steps 30 --warmup 5 --no-regimes --no-cpu-baseline --no-live-pmc --turn-frames 0 > $O/${W}_thin.json 2> $O/${W}_thin.err; echo "$W thin rc $?"
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s2i/*.json')):
    t=open(f).read(); i=t.find('{"metric')
    if i<0: print(f,'no line'); continue
    d=json.loads(t[i:].splitlines()[0])
    k=lambda l: (d[l]['kernel_choice'] or {}).get('kept')
    s=d['serial']
    print(f.split('/')[-1], 'serial kernel', s['kernel_ms_median'], s['kernel_ms_p10_p90'], 'ms', s['ms_per_step'], 'Gs', s['value'], '| pipelined', d['pipelined_one_frame_per_launch']['ms_per_step'], d['pipelined_one_frame_per_launch']['value'], '| batched', d['overlapped']['ms_per_step'], d['overlapped']['value'], '| kept', k('serial'), k('pipelined_one_frame_per_launch'), k('overlapped'), '| traffic', d['roofline'].get('traffic'), 'frac', d['roofline'].get('frac'))
P
