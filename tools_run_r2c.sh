mkdir -p gpurun_out/r2c
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r2c/tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r2c/tests.log
if [ $rc -ge 100 ]; then exit 1; fi
for L in 0 1; do
  timeout -k 10 120 python bench.py --steps 30 --warmup 5 --air noisy --no-regimes --no-cpu-baseline --pmc-extra --layout $L > gpurun_out/r2c/noisy_L$L.json 2> gpurun_out/r2c/noisy_L$L.err; echo "noisy L$L rc $?"
  timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-regimes --no-cpu-baseline --pmc-extra --layout $L > gpurun_out/r2c/def_L$L.json 2> gpurun_out/r2c/def_L$L.err; echo "def L$L rc $?"
  for T in default thin; do
    timeout -k 10 120 python bench.py --workload C2 --tf $T --steps 30 --warmup 5 --no-cpu-baseline --no-live-pmc --layout $L > gpurun_out/r2c/c2_${T}_L$L.json 2> gpurun_out/r2c/c2_${T}_L$L.err; echo "c2 $T L$L rc $?"
    timeout -k 10 120 python bench.py --workload C2 --air noisy --tf $T --steps 30 --warmup 5 --no-cpu-baseline --no-live-pmc --layout $L > gpurun_out/r2c/c2n_${T}_L$L.json 2> gpurun_out/r2c/c2n_${T}_L$L.err; echo "c2n $T L$L rc $?"
  done
  timeout -k 10 120 python bench.py --workload C4 --steps 30 --warmup 5 --no-cpu-baseline --no-live-pmc --layout $L > gpurun_out/r2c/c4_L$L.json 2> gpurun_out/r2c/c4_L$L.err; echo "c4 L$L rc $?"
done
