#!/bin/bash
# the N > 1 bookkeeping of bench.py after this round's changes: two self-launched ranks on one GPU (gloo rehearsal), and the
# real RCCL loop with a world of one
O=gpurun_out/r4o
mkdir -p $O
VR_BENCH_DEVICE=0 VR_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > $O/gloo2.json 2> $O/gloo2.err || { tail -8 $O/gloo2.err; exit 1; }
VR_BENCH_SELF_GATHER=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes --no-live-pmc > $O/selfgather.json 2> $O/selfgather.err || { tail -8 $O/selfgather.err; exit 1; }
python - <<'PY'
import json
for n in ('gloo2','selfgather'):
    d=json.load(open(f'gpurun_out/r4o/{n}.json'))
    c=d['config']
    print(n, d['n_gpus'], d['value'], d['ms_per_step'], c.get('launched_by'), c.get('frame_equals_single_rank_render'), c.get('rccl_nranks'), [r and r.get('device') for r in c.get('ranks',[])])
    print('   timeline', json.dumps(d.get('rank0_stage_timeline'))[:400])
PY
