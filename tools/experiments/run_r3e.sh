#!/bin/bash
# round 3: the new bench.py (turntable, serial headline, present leg, self-launcher) and the multi-GPU loop's new outputs
set -x
O=gpurun_out/r3e
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_mgpu_loopback_gpu.py tests/test_mgpu_gpu.py tests/test_parity_gpu.py -x -q -k "mgpu or ranks or present or frames_per_launch or pipelined or kernel_times or abi" > $O/pytest_mgpu.txt 2>&1
rc=$?
tail -5 $O/pytest_mgpu.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
VR_BENCH_SELF_GATHER=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_selfgather.json 2> $O/bench_selfgather.err || { tail -20 $O/bench_selfgather.err; exit 1; }
VR_BENCH_SELF_GATHER=1 VR_MGPU_EXP_SHARE=8 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_share8.json 2> $O/bench_share8.err || { tail -20 $O/bench_share8.err; exit 1; }
VR_BENCH_DEVICE=0 VR_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err || { tail -20 $O/bench_gloo2.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e/bench_*.json')):
    try: d=json.load(open(f))
    except Exception as e:
        print(f,'unreadable',e); continue
    print(f.split('/')[-1], 'value', d['value'], 'ms', d['ms_per_step'], 'present', (d.get('serial_with_present') or {}).get('ms_per_step'), 'pipe', (d.get('pipelined_one_frame_per_launch') or {}).get('ms_per_step'), 'batched', d['overlapped']['ms_per_step'], d['overlapped']['value'])
    print('   ', {k:d['config'].get(k) for k in ('rccl_nranks','frame_equals_single_rank_render','frames_compared','launched_by','composited_samples_per_frame_min_max')})
    if 'rank0_stage_timeline' in d: print('   ', json.dumps(d['rank0_stage_timeline'])[:900])
    r=d['roofline']
    print('   roofline', r.get('frac'), r.get('traffic'), json.dumps(r.get('turntable_vs_identical'))[:600])
PY
