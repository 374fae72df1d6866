#!/bin/bash
# The unlit shader's no-skip form (flavour 16, 93 VGPRs) with 8 / 12 / 16 wavefronts per CU: C2 with noisy air (nothing to skip)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2k
mkdir -p $O
cd $R
for T in 512 768 1024; do
  VR_EXP_P2_THREADS=$T python3 bench.py --workload C2 --air noisy --flavour 16 --steps 40 --warmup 8 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 > $O/c2_noisy_$T.json 2> $O/c2_noisy_$T.err
  python3 - <<P
import json
t=open('$O/c2_noisy_$T.json').read(); d=json.loads(t[t.find('{"metric'):].splitlines()[0])
print('C2 noisy f16 threads $T: serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'pipelined', d['pipelined_one_frame_per_launch']['ms_per_step'], 'batched', d['overlapped']['ms_per_step'], 'ran', d['serial']['kernel_choice']['ran_last'])
P
done
python3 bench.py --workload C2 --air noisy --steps 40 --warmup 8 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 > $O/c2_noisy_default.json 2> $O/c2_noisy_default.err
python3 - <<P
import json
t=open('$O/c2_noisy_default.json').read(); d=json.loads(t[t.find('{"metric'):].splitlines()[0])
print('C2 noisy default: serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_choice'], 'pipelined', d['pipelined_one_frame_per_launch']['ms_per_step'], 'batched', d['overlapped']['ms_per_step'])
P
