#!/bin/bash
# round 4: what a step slot of march_p2_kernel costs by kind (-DVR_P2_DEBUG=1 build, tools/block_trace.py's regression)
O=gpurun_out/r4v
mkdir -p $O
VR_EXTRA_HIPCC_FLAGS="-DVR_P2_DEBUG=1" python -c "from volumerendering_amd import build as b; b.build_hip()" > $O/build_dbg.txt 2>&1 || { tail -5 $O/build_dbg.txt; exit 1; }
VR_P2_DEBUG=1 timeout -k 10 200 python tools/block_trace.py --flavour 17 > $O/trace_c3.txt 2>&1; grep "slot costs\|slot totals\|packet time\|loop counters\|device span" $O/trace_c3.txt
VR_P2_DEBUG=1 timeout -k 10 200 python tools/block_trace.py --flavour 17 --tf thin > $O/trace_thin.txt 2>&1; grep "slot costs\|slot totals\|packet time\|device span" $O/trace_thin.txt
VR_P2_DEBUG=1 timeout -k 10 200 python tools/block_trace.py --flavour 17 --workload C4 > $O/trace_c4.txt 2>&1; grep "slot costs\|slot totals\|packet time\|device span" $O/trace_c4.txt
