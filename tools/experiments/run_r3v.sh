#!/bin/bash
set -x
O=gpurun_out/r3v
mkdir -p $O
for fl in 6 17 12; do
  timeout -k 10 300 python tools/block_trace.py --flavour $fl > $O/trace_c3_f$fl.txt 2>&1 || { tail -5 $O/trace_c3_f$fl.txt; exit 1; }
done
timeout -k 10 300 python tools/block_trace.py --flavour 17 --tf thin > $O/trace_thin_f17.txt 2>&1
timeout -k 10 300 python tools/block_trace.py --flavour 12 --tf thin > $O/trace_thin_f12.txt 2>&1
tail -12 $O/trace_c3_f17.txt
