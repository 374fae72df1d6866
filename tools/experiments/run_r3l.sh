#!/bin/bash
# round 3: storage brick edge of the bricked layout: 2, 4 (default), 8 voxels.  Rebuilds libvr_hip.so on the box per size.
set -x
O=gpurun_out/r3l
mkdir -p $O
for S in 3 1 2; do
  VR_EXTRA_HIPCC_FLAGS="-DVR_VOX_BRICK_SHIFT=$S" python -c "from volumerendering_amd import build as b; b.build_hip(force=True); b.build_host(force=True); b.build_mgpu(force=True)" > $O/build_s$S.txt 2>&1 || { tail -5 $O/build_s$S.txt; exit 1; }
  timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -k "every_variant_bit_exact or hostile or persistent_wavefronts_queue" > $O/pytest_s$S.txt 2>&1 || { tail -5 $O/pytest_s$S.txt; exit 1; }
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-regimes --no-live-pmc > $O/bench_c3_s$S.json 2> $O/bench_c3_s$S.err || { tail -5 $O/bench_c3_s$S.err; exit 1; }
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --air noisy --no-cpu-baseline --no-regimes > $O/bench_noisy_s$S.json 2> $O/bench_noisy_s$S.err || { tail -5 $O/bench_noisy_s$S.err; exit 1; }
  timeout -k 10 300 python bench.py --workload C2 --steps 40 --warmup 8 --no-cpu-baseline --no-live-pmc > $O/bench_C2_s$S.json 2> $O/bench_C2_s$S.err || { tail -5 $O/bench_C2_s$S.err; exit 1; }
  timeout -k 10 300 python bench.py --workload C4 --steps 30 --warmup 8 --no-cpu-baseline --no-live-pmc > $O/bench_C4_s$S.json 2> $O/bench_C4_s$S.err || { tail -5 $O/bench_C4_s$S.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3l/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'value', d['value'], 'serial', d['serial']['ms_per_step'], d['serial']['kernel_ms_median'], d['serial']['kernel_ms_p10_p90'], 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'], 'fl', d['config']['kernel_flavour_resolved'], 'GB', round((r.get('traffic') or 0)/1e9,2))
PY
