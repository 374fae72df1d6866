# usage: tools/run_exp.sh <outdir> ; runs a list of bench experiments given on stdin: "<tag> | <env assignments> | <bench args>"
OUT=gpurun_out/$1; mkdir -p $OUT
while IFS='|' read -r TAG ENVS ARGS; do
  TAG=$(echo $TAG); [ -z "$TAG" ] && continue
  env $ENVS timeout -k 10 150 python bench.py --steps 30 --warmup 5 --no-regimes --no-cpu-baseline $ARGS > $OUT/$TAG.json 2> $OUT/$TAG.err; echo "$TAG rc $?"
done
