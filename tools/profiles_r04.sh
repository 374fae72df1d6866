#!/bin/bash
# Profiling recipe of round 4 (run on the GPU box via gpurun); tools/collect_profiles.py 04 copies the summaries into profiles/r04_*.
#   1. rocprofv3 --kernel-trace --stats of the driver's bench command (the march kernel's average must agree with the line's
#      HIP-event median) and of the same with every leg one frame at a time
#   2. bench lines with every PMC counter (--pmc-extra) for C3: the default (measured choice), forced kernels, regimes
#   3. the other BASELINE configurations, thin tables, the rank-share rehearsals (C3 / C5, world 2 / 4 / 8)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --no-regimes --steps 20 --warmup 5 > $OUT/trace_bench.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; }
echo "trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --no-regimes --steps 40 --warmup 5 --in-flight 1 --turn-frames 0 > $OUT/trace_serial_bench.json 2> $OUT/trace_serial.err || { echo "serial trace failed"; tail -5 $OUT/trace_serial.err; }
echo "serial trace done"
cd $R
B="python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-regimes --pmc-extra"
$B > $OUT/c3_default.json 2> $OUT/c3_default.err; echo "c3 default rc $?"
$B --flavour 6 --turn-frames 0 > $OUT/c3_f6.json 2> $OUT/c3_f6.err; echo "c3 f6 rc $?"
$B --flavour 17 --turn-frames 0 > $OUT/c3_f17.json 2> $OUT/c3_f17.err; echo "c3 f17 rc $?"
$B --arith fused --turn-frames 0 > $OUT/c3_fused.json 2> $OUT/c3_fused.err; echo "c3 fused rc $?"
$B --tf thin > $OUT/c3_thin.json 2> $OUT/c3_thin.err; echo "c3 thin rc $?"
$B --air noisy > $OUT/c3_noisy.json 2> $OUT/c3_noisy.err; echo "c3 noisy rc $?"
$B --air noisy --flavour 6 --turn-frames 0 > $OUT/c3_noisy_f6.json 2> $OUT/c3_noisy_f6.err; echo "c3 noisy f6 rc $?"
for W in C1 C2 C4 C5; do
  python3 bench.py --workload $W --steps 30 --warmup 5 --no-regimes --no-cpu-baseline > $OUT/${W}_default.json 2> $OUT/${W}_default.err; echo "$W rc $?"
  python3 bench.py --workload $W --tf thin --steps 30 --warmup 5 --no-regimes --no-cpu-baseline --no-live-pmc --turn-frames 0 > $OUT/${W}_thin.json 2> $OUT/${W}_thin.err; echo "$W thin rc $?"
done
python3 bench.py --workload C4 --flavour 6 --steps 30 --warmup 5 --no-regimes --no-cpu-baseline --no-live-pmc --turn-frames 0 > $OUT/C4_f6.json 2> $OUT/C4_f6.err; echo "C4 f6 rc $?"
python3 bench.py --workload C5 --flavour 17 --steps 30 --warmup 5 --no-regimes --no-cpu-baseline --no-live-pmc --turn-frames 0 > $OUT/C5_f17.json 2> $OUT/C5_f17.err; echo "C5 f17 rc $?"
for N in 2 4 8; do
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 > $OUT/c3_share${N}.json 2> $OUT/c3_share${N}.err; echo "share $N rc $?"
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N python3 bench.py --workload C5 --steps 20 --warmup 4 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 > $OUT/C5_share${N}.json 2> $OUT/C5_share${N}.err; echo "C5 share $N rc $?"
done
VR_BENCH_SELF_GATHER=1 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-regimes --no-live-pmc --turn-frames 0 > $OUT/c3_selfgather.json 2> $OUT/c3_selfgather.err; echo "selfgather rc $?"
python3 tools/block_trace.py --flavour 17 > $OUT/block_trace_c3_f17.txt 2>&1
python3 tools/block_trace.py --flavour 6 > $OUT/block_trace_c3_f6.txt 2>&1
python3 tools/block_trace.py --flavour 16 --air noisy > $OUT/block_trace_noisy_f16.txt 2>&1
python3 tools/block_trace.py --flavour 17 --workload C4 > $OUT/block_trace_c4_f17.txt 2>&1
./tools/ubench/struct_buffer > $OUT/struct_buffer.txt 2>&1
# (hipcc -O3 --offload-arch=gfx950 -o tools/ubench/stream_gap tools/ubench/stream_gap.hip) what a stream pays between two launches
./tools/ubench/stream_gap > $OUT/stream_gap.txt 2>&1
# the serial leg over 200 cameras: the last launch's packets, and from a kernel trace the idle time between two march kernels
python3 tools/launch_gap.py --run > $OUT/launch_gap_c3_f17.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/launch_gap_kt -- python3 $R/tools/launch_gap.py --run > /dev/null 2>&1 && python3 $R/tools/launch_gap.py --read $OUT/launch_gap_kt >> $OUT/launch_gap_c3_f17.txt 2>&1; rm -rf $OUT/launch_gap_kt)
G="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS;SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC;SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT;SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH"
python3 tools/pmc_probe.py --flavour 17 --groups "$G" > $OUT/pmc_probe_c3_f17.json 2> $OUT/pmc_probe_c3_f17.err
python3 tools/pmc_probe.py --flavour 6 --groups "$G" > $OUT/pmc_probe_c3_f6.json 2> $OUT/pmc_probe_c3_f6.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc $?"
grep "march" $OUT/trace/*/*kernel_stats.csv | head -8
echo done
