#!/bin/bash
# Host-side waits for the sorts, one frame at a time only: the GPU suite, the driver's command (the final profile), C1 / C2 / C4 / noisy lines
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2p
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
for W in C1 C2 C4; do
  python3 bench.py --workload $W --steps 30 --warmup 5 --no-regimes --no-cpu-baseline --no-live-pmc --turn-frames 0 > $O/$W.json 2> $O/$W.err; echo "$W rc $?"
done
python3 bench.py --air noisy --steps 20 --warmup 5 --no-regimes --no-cpu-baseline --no-live-pmc --turn-frames 0 > $O/c3_noisy.json 2> $O/c3_noisy.err
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s2p/*.json')):
    t=open(f).read(); i=t.find('{"metric')
    if i<0: print(f,'no line'); continue
    d=json.loads(t[i:].splitlines()[0])
    s=d['serial']
    print(f.split('/')[-1], 'value', d['value'], 'serial', s['ms_per_step'], s['kernel_ms_median'], 'present', d['serial_with_present']['ms_per_step'], 'pipelined', d['pipelined_one_frame_per_launch']['ms_per_step'], 'batched', d['overlapped']['ms_per_step'], 'sync', d['sync_8d']['t_frame_ms_median'], 'turn', (d.get('full_turn') or {}).get('ms_per_frame_mean'), 'kept', s['kernel_choice']['kept'], 'frac', d['roofline'].get('frac'), 'parity', (d.get('parity') or {}).get('bit_equal'))
P
