#!/bin/bash
# debug build of the two-steps-ahead kernel: the loop's own counters per packet
set -x
O=gpurun_out/r4e
mkdir -p $O
VR_P2_DEBUG=1 timeout -k 10 300 python tools/block_trace.py --flavour 17 > $O/trace_c3_f17_dbg.txt 2>&1
grep "loop counters" $O/trace_c3_f17_dbg.txt
VR_P2_DEBUG=1 timeout -k 10 300 python tools/block_trace.py --flavour 17 --tf thin > $O/trace_thin_f17_dbg.txt 2>&1
grep "loop counters" $O/trace_thin_f17_dbg.txt
