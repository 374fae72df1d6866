#!/bin/bash
# Profiling recipe of round 2 (run on the GPU box via gpurun): the summaries copied into profiles/r02_* come from here.
#   1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel average durations)
#   2. bench lines with every PMC counter (--pmc-extra: rocprofv3 --pmc passes, one counter group per run, launched by
#      bench.py itself on the same scene) for the regimes and A/B modes DESIGN.md quotes
#   3. bench lines of the other BASELINE configurations
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --no-regimes --steps 20 --warmup 3 > $OUT/trace_bench.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; }
echo "trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --no-regimes --steps 40 --warmup 5 --in-flight 1 > $OUT/trace_serial_bench.json 2> $OUT/trace_serial.err || { echo "serial trace failed"; tail -5 $OUT/trace_serial.err; }
echo "serial trace done"
cd $R
B="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-regimes --pmc-extra"
$B > $OUT/c3_default.json 2> $OUT/c3_default.err; echo "c3 default rc $?"
$B --arith fused > $OUT/c3_fused.json 2> $OUT/c3_fused.err; echo "c3 fused rc $?"
$B --air noisy > $OUT/c3_noisy.json 2> $OUT/c3_noisy.err; echo "c3 noisy rc $?"
$B --air noisy --layout 1 > $OUT/c3_noisy_vec4.json 2> $OUT/c3_noisy_vec4.err; echo "c3 noisy vec4 rc $?"
$B --air noisy --layout 2 > $OUT/c3_noisy_otf.json 2> $OUT/c3_noisy_otf.err; echo "c3 noisy otf rc $?"
$B --layout 2 > $OUT/c3_otf.json 2> $OUT/c3_otf.err; echo "c3 otf rc $?"
$B --tf thin > $OUT/c3_thin.json 2> $OUT/c3_thin.err; echo "c3 thin rc $?"
$B --flavour 1 > $OUT/c3_flavour1.json 2> $OUT/c3_flavour1.err; echo "c3 flavour1 rc $?"
$B --air noisy --flavour 2 > $OUT/c3_noisy_wtb.json 2> $OUT/c3_noisy_wtb.err; echo "c3 noisy wtb rc $?"
VR_EXP_WAVES_PER_BLOCK=4 $B > $OUT/c3_wpb4.json 2> $OUT/c3_wpb4.err; echo "c3 wpb4 rc $?"
VR_EXP_ORDER=0 $B > $OUT/c3_noorder.json 2> $OUT/c3_noorder.err; echo "c3 noorder rc $?"
for W in C1 C2 C4 C5; do
  python3 bench.py --workload $W --steps 50 --warmup 5 --no-regimes > $OUT/${W}_default.json 2> $OUT/${W}_default.err; echo "$W rc $?"
  python3 bench.py --workload $W --tf thin --steps 50 --warmup 5 --no-regimes --no-cpu-baseline --no-live-pmc > $OUT/${W}_thin.json 2> $OUT/${W}_thin.err; echo "$W thin rc $?"
done
VR_BENCH_SELF_GATHER=1 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-regimes --no-live-pmc > $OUT/c3_selfgather.json 2> $OUT/c3_selfgather.err; echo "selfgather rc $?"
# rank 0's timeline of an N-rank run, rehearsed on this one GPU (VR_MGPU_EXP_SHARE: its share of the tiles, the gather of its
# segment through RCCL, the un-permute of whole frames); steady state (400 frames) and the driver's 20-frame run
for N in 2 4 8; do
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-regimes --no-live-pmc > $OUT/c3_share${N}.json 2> $OUT/c3_share${N}.err; echo "share $N rc $?"
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc > $OUT/c3_share${N}_k20.json 2> $OUT/c3_share${N}_k20.err; echo "share $N k20 rc $?"
done
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc $?"
echo done
