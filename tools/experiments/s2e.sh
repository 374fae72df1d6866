#!/bin/bash
# The heaviest packets alone with one wavefront on their SIMD (VR_EXP_P2_LONE=n workgroups per class) on top of the work-sorted
# launch order (VR_EXP_ORDER_KEY=1): wall time per frame of the serial leg (tools/launch_gap.py --run), C3, flavour 17
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2e
mkdir -p $O
cd $R
VR_EXP_ORDER_KEY=1 VR_EXP_P2_LONE=2 timeout -k 10 300 python3 -m pytest tests/test_p2_gpu.py -x -q > $O/pytest_p2_lone2.txt 2>&1 || { tail -20 $O/pytest_p2_lone2.txt; exit 1; }
tail -2 $O/pytest_p2_lone2.txt
for C in "0 0" "1 0" "1 1" "1 2" "1 4" "1 8" "0 2" "0 4"; do set -- $C
  VR_EXP_ORDER_KEY=$1 VR_EXP_P2_LONE=$2 timeout -k 10 120 python3 tools/launch_gap.py --run > $O/key$1_lone$2.txt 2>&1 || { tail -5 $O/key$1_lone$2.txt; exit 1; }
  echo "== key $1 lone $2"; grep -v amdgpu.ids $O/key$1_lone$2.txt
done
