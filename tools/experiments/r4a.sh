#!/bin/bash
# round 4, first probe: (1) indexed buffer loads beyond 4 GiB, (2) box calibration with round 3's library, (3) what share of
# the sampled steps of march_p2_kernel is shaded (VR_P2_DEBUG build)
O=gpurun_out/r4a
mkdir -p $O
timeout -k 10 120 ./tools/ubench/struct_buffer > $O/struct_buffer.txt 2>&1; echo "struct_buffer rc $?"; cat $O/struct_buffer.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3.json 2> $O/bench_c3.err || { tail -5 $O/bench_c3.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4a/bench_c3.json'))
for k in ('serial','pipelined_one_frame_per_launch','overlapped'):
    s=d[k]; print(k, s['ms_per_step'], s['kernel_ms_median'], s['value'])
PY
VR_EXTRA_HIPCC_FLAGS="-DVR_P2_DEBUG=1" python -c "from volumerendering_amd import build as b; b.build_hip()" > $O/build_dbg.txt 2>&1 || { tail -5 $O/build_dbg.txt; exit 1; }
VR_P2_DEBUG=1 timeout -k 10 200 python tools/block_trace.py --flavour 17 > $O/trace_dbg_c3.txt 2>&1; grep "loop counters" $O/trace_dbg_c3.txt
VR_P2_DEBUG=1 timeout -k 10 200 python tools/block_trace.py --flavour 17 --tf thin > $O/trace_dbg_thin.txt 2>&1; grep "loop counters" $O/trace_dbg_thin.txt
echo done
