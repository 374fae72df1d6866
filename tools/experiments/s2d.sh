#!/bin/bash
# The launch order sorted by the packets' counted work (VR_EXP_ORDER_KEY=1) x raised issue priority for the first n packets of every
# class's order (VR_EXP_P2_PRIO=n): wall time per frame of the serial leg (tools/launch_gap.py --run), C3, flavour 17
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2d
mkdir -p $O
cd $R
for K in 0 1; do for P in 0 32 128; do
  VR_EXP_ORDER_KEY=$K VR_EXP_P2_PRIO=$P python3 tools/launch_gap.py --run > $O/key${K}_prio${P}.txt 2>&1
  echo "== key $K prio $P"; grep -v amdgpu.ids $O/key${K}_prio${P}.txt
done; done
VR_EXP_ORDER_KEY=1 timeout -k 10 300 python3 -m pytest tests/test_p2_gpu.py -x -q > $O/pytest_p2_key1.txt 2>&1; tail -3 $O/pytest_p2_key1.txt
