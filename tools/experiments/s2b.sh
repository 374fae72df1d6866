#!/bin/bash
# What the serial leg pays between two march kernels: kernel trace of tools/launch_gap.py with and without the waits for
# sorts that finished long ago (VR_EXP_ELIDE_WAITS)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for E in 0 1; do
  export VR_EXP_ELIDE_WAITS=$E
  python3 $R/tools/launch_gap.py --run > $O/run_plain_$E.txt 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/kt$E -- python3 $R/tools/launch_gap.py --run > $O/run_$E.txt 2>&1 || { tail -5 $O/run_$E.txt; exit 1; }
  python3 $R/tools/launch_gap.py --read $O/kt$E > $O/gap_$E.txt 2>&1
  rm -rf $O/kt$E
done
tail -n 20 $O/run_plain_*.txt $O/gap_*.txt
