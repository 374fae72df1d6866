#!/bin/bash
# where march_p2_kernel waits: -DVR_P2_DEBUG=2 build (shader-clock reads around the corner wait and the byte wait)
O=gpurun_out/r4k
mkdir -p $O
VR_EXTRA_HIPCC_FLAGS="-DVR_P2_DEBUG=2" python -c "from volumerendering_amd import build as b; b.build_hip()" > $O/build_dbg.txt 2>&1 || { tail -5 $O/build_dbg.txt; exit 1; }
VR_P2_DEBUG=2 timeout -k 10 200 python tools/block_trace.py --flavour 17 > $O/trace_c3.txt 2>&1; grep "wait cycles\|loop counters\|device span" $O/trace_c3.txt
VR_P2_DEBUG=2 timeout -k 10 200 python tools/block_trace.py --flavour 16 --air noisy > $O/trace_noisy.txt 2>&1; grep "wait cycles\|loop counters\|device span" $O/trace_noisy.txt
VR_P2_DEBUG=2 timeout -k 10 200 python tools/block_trace.py --flavour 17 --workload C4 > $O/trace_c4.txt 2>&1; grep "wait cycles\|loop counters\|device span" $O/trace_c4.txt
