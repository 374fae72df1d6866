#!/bin/bash
# Second profiling pass of round 3 (after flavours 16 / 17 became the defaults of their regimes): refreshes the summaries of
# tools/profiles_r03.sh that the new defaults change, into the same directory (tools/collect_profiles.py 03 copies them).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --no-regimes --steps 20 --warmup 5 > $OUT/trace_bench.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; }
echo "trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --no-regimes --steps 40 --warmup 5 --in-flight 1 > $OUT/trace_serial_bench.json 2> $OUT/trace_serial.err || { echo "serial trace failed"; tail -5 $OUT/trace_serial.err; }
echo "serial trace done"
cd $R
B="python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-regimes --pmc-extra"
$B > $OUT/c3_default.json 2> $OUT/c3_default.err; echo "c3 default rc $?"
$B --flavour 6 > $OUT/c3_f6.json 2> $OUT/c3_f6.err; echo "c3 f6 rc $?"
$B --arith fused > $OUT/c3_fused.json 2> $OUT/c3_fused.err; echo "c3 fused rc $?"
$B --identical-frames > $OUT/c3_identical.json 2> $OUT/c3_identical.err; echo "c3 identical rc $?"
$B --air noisy > $OUT/c3_noisy.json 2> $OUT/c3_noisy.err; echo "c3 noisy rc $?"
$B --air noisy --flavour 13 > $OUT/c3_noisy_f13.json 2> $OUT/c3_noisy_f13.err; echo "c3 noisy f13 rc $?"
$B --air noisy --arith fused > $OUT/c3_noisy_fused.json 2> $OUT/c3_noisy_fused.err; echo "c3 noisy fused rc $?"
$B --tf thin --flavour 17 > $OUT/c3_thin_f17.json 2> $OUT/c3_thin_f17.err; echo "c3 thin f17 rc $?"
python3 tools/block_trace.py --flavour 17 > $OUT/block_trace_c3_f17.txt 2>&1
python3 tools/block_trace.py --flavour 6 > $OUT/block_trace_c3_f6.txt 2>&1
python3 tools/block_trace.py --flavour 16 --air noisy > $OUT/block_trace_noisy_f16.txt 2>&1
./tools/ubench/valu_issue > $OUT/valu_issue.txt 2>&1
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/ubench_pmc -- $R/tools/ubench/valu_issue > $OUT/ubench_pmc.txt 2>&1
cd $R
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc $?"
grep "march" $OUT/trace/*/*kernel_stats.csv | head -8
echo done
