#!/bin/bash
# march_p2_kernel of the unlit shader with 16 wavefronts per CU (101 VGPRs: 4 per SIMD fit) against 12: C2 default and thin table, forced 17
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2h
mkdir -p $O
cd $R
for T in 768 1024; do for TF in default thin; do
  VR_EXP_P2_THREADS=$T python3 bench.py --workload C2 --tf $TF --flavour 17 --steps 40 --warmup 8 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 > $O/c2_${TF}_$T.json 2> $O/c2_${TF}_$T.err
  python3 - <<P
import json
t=open('$O/c2_${TF}_$T.json').read(); d=json.loads(t[t.find('{"metric'):].splitlines()[0])
print('C2 $TF threads $T: serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], 'pipelined', d['pipelined_one_frame_per_launch']['ms_per_step'], 'batched', d['overlapped']['ms_per_step'], 'ran', d['serial']['kernel_choice']['ran_last'])
P
done; done
VR_EXP_P2_THREADS=1024 timeout -k 10 300 python3 -m pytest tests/test_p2_gpu.py tests/test_parity_gpu.py -x -q -k "not lds_tiles" > $O/pytest_1024.txt 2>&1; tail -2 $O/pytest_1024.txt
