#!/bin/bash
# Profiling recipe of round 3 (run on the GPU box via gpurun): the summaries copied into profiles/r03_* come from here
# (tools/collect_profiles.py 03).
#   1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel average durations; the march kernel's average
#      must agree with the bench line's HIP-event median)
#   2. bench lines with every PMC counter (--pmc-extra: rocprofv3 --pmc passes, one counter group per run, launched by bench.py
#      itself on the same scene and turntable) for the regimes, layouts and kernel forms DESIGN.md quotes
#   3. bench lines of the other BASELINE configurations, the multi-rank rehearsals (C3 and C5 shares), the microbenchmark
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --no-regimes --steps 20 --warmup 5 > $OUT/trace_bench.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; }
echo "trace done"
cd $R
B="python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-regimes --pmc-extra"
$B > $OUT/c3_default.json 2> $OUT/c3_default.err; echo "c3 default rc $?"
$B --layout 3 > $OUT/c3_layout3.json 2> $OUT/c3_layout3.err; echo "c3 layout3 rc $?"
$B --arith fused > $OUT/c3_fused.json 2> $OUT/c3_fused.err; echo "c3 fused rc $?"
$B --identical-frames > $OUT/c3_identical.json 2> $OUT/c3_identical.err; echo "c3 identical rc $?"
$B --flavour 12 > $OUT/c3_f12.json 2> $OUT/c3_f12.err; echo "c3 f12 rc $?"
$B --flavour 13 > $OUT/c3_f13.json 2> $OUT/c3_f13.err; echo "c3 f13 rc $?"
# (flavours 14 and 15 need the experimental build: VR_EXPERIMENTAL_FLAVOURS=1 python -c "from volumerendering_amd import build; build.build_all()")
$B --flavour 14 > $OUT/c3_f14.json 2> $OUT/c3_f14.err; echo "c3 f14 rc $?"
$B --flavour 15 > $OUT/c3_f15.json 2> $OUT/c3_f15.err; echo "c3 f15 rc $?"
$B --tf thin > $OUT/c3_thin.json 2> $OUT/c3_thin.err; echo "c3 thin rc $?"
$B --air noisy > $OUT/c3_noisy.json 2> $OUT/c3_noisy.err; echo "c3 noisy rc $?"
$B --air noisy --flavour 6 > $OUT/c3_noisy_f6.json 2> $OUT/c3_noisy_f6.err; echo "c3 noisy f6 rc $?"
$B --air noisy --flavour 6 --layout 3 > $OUT/c3_noisy_f6_layout3.json 2> $OUT/c3_noisy_f6_layout3.err; echo "c3 noisy f6 layout3 rc $?"
$B --air noisy --flavour 15 > $OUT/c3_noisy_f15.json 2> $OUT/c3_noisy_f15.err; echo "c3 noisy f15 rc $?"
$B --air noisy --flavour 2 > $OUT/c3_noisy_wtb.json 2> $OUT/c3_noisy_wtb.err; echo "c3 noisy wtb rc $?"
$B --air noisy --arith fused > $OUT/c3_noisy_fused.json 2> $OUT/c3_noisy_fused.err; echo "c3 noisy fused rc $?"
for W in C1 C2 C4 C5; do
  python3 bench.py --workload $W --steps 30 --warmup 5 --no-regimes --no-cpu-baseline > $OUT/${W}_default.json 2> $OUT/${W}_default.err; echo "$W rc $?"
  python3 bench.py --workload $W --tf thin --steps 30 --warmup 5 --no-regimes --no-cpu-baseline --no-live-pmc > $OUT/${W}_thin.json 2> $OUT/${W}_thin.err; echo "$W thin rc $?"
done
VR_BENCH_SELF_GATHER=1 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-regimes --no-live-pmc > $OUT/c3_selfgather.json 2> $OUT/c3_selfgather.err; echo "selfgather rc $?"
# rank 0's timeline of an N-rank run, rehearsed on this one GPU (VR_MGPU_EXP_SHARE: its share of the tiles, the gather of its
# segment through RCCL, the root's output pass over whole frames); C3 and C5
for N in 2 4 8; do
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-regimes --no-live-pmc > $OUT/c3_share${N}.json 2> $OUT/c3_share${N}.err; echo "share $N rc $?"
  VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=$N python3 bench.py --workload C5 --steps 20 --warmup 4 --no-cpu-baseline --no-regimes --no-live-pmc > $OUT/C5_share${N}.json 2> $OUT/C5_share${N}.err; echo "C5 share $N rc $?"
done
VR_BENCH_DEVICE=0 VR_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/c3_gloo2_selflaunched.json 2> $OUT/c3_gloo2_selflaunched.err; echo "gloo2 rc $?"
python3 tools/block_trace.py --flavour 6 > $OUT/block_trace_c3_f6.txt 2>&1
python3 tools/block_trace.py --flavour 13 > $OUT/block_trace_c3_f13.txt 2>&1
python3 tools/block_trace.py --flavour 15 > $OUT/block_trace_c3_f15.txt 2>&1
./tools/ubench/valu_issue > $OUT/valu_issue.txt 2>&1
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/ubench_pmc -- $R/tools/ubench/valu_issue > $OUT/ubench_pmc.txt 2>&1
cd $R
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc $?"
echo done
