#!/bin/bash
# The launch order's sort key (chain / the packet's own time) x CUs left free for the sort of the launch before: wall time per frame
# of the serial leg (tools/launch_gap.py --run), C3, flavour 17; then the full 20-frame and 100-frame bench legs for the best
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2c
mkdir -p $O
cd $R
for K in 0 1; do for S in 0 1 8; do
  VR_EXP_ORDER_KEY=$K VR_EXP_P2_SPARE=$S python3 tools/launch_gap.py --run > $O/key${K}_spare${S}.txt 2>&1
  echo "== key $K spare $S"; grep -v amdgpu.ids $O/key${K}_spare${S}.txt
done; done
