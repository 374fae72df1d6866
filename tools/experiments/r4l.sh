#!/bin/bash
O=gpurun_out/r4l
mkdir -p $O
VR_EXTRA_HIPCC_FLAGS="-DVR_P2_DEBUG=3" python -c "from volumerendering_amd import build as b; b.build_hip()" > $O/build_dbg.txt 2>&1 || { tail -5 $O/build_dbg.txt; exit 1; }
VR_P2_DEBUG=3 timeout -k 10 200 python tools/block_trace.py --flavour 17 > $O/trace_c3.txt 2>&1; grep "packet cycles\|device span" $O/trace_c3.txt
VR_P2_DEBUG=3 timeout -k 10 200 python tools/block_trace.py --flavour 16 --air noisy > $O/trace_noisy.txt 2>&1; grep "packet cycles\|device span" $O/trace_noisy.txt
