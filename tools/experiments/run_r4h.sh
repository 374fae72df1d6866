#!/bin/bash
# the whole GPU suite with the new default, then the driver's bench command
set -x
O=gpurun_out/r4h
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
rc=$?
tail -5 $O/pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4h/bench_default.json'))
print({k:d[k] for k in ('metric','value','ms_per_step','fps')}, d['config'].get('kernel_flavour_resolved'))
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','traffic','kernel','kernel_ms')})
for leg in ('serial','serial_with_present','pipelined_one_frame_per_launch','overlapped'):
    print(leg, {k:v for k,v in d[leg].items() if k in ('ms_per_step','value','kernel_ms_median','frames_per_launch','launches_in_flight')})
print('cpu', d['cpu_baseline'])
PY
