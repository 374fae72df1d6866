#!/bin/bash
O=gpurun_out/r4n
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; tail -4 $O/pytest.txt; [ $rc -ne 0 ] && exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time python bench.py > $O/bench_noargs.json 2> $O/bench_noargs.err ) 2> $O/bench_noargs.time; tail -3 $O/bench_noargs.time
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4n/bench_noargs.json'))
print(d['value'], d['ms_per_step'], d['steps'], d['roofline']['frac'], d['parity']['bit_equal'], d['full_turn']['ms_per_frame_mean'], d['sync_8d']['t_frame_ms_median'])
PY
