#!/bin/bash
# dev: per-operation table under different values of ONE environment variable, interleaved on one device:
#   bash tools/ab_env.sh CZC_GEMM_STAGGER "off 4:6:1024 2:12:1024" 'x768 '
set -e
mkdir -p gpurun_out
VAR=$1
for r in 1 2; do
for v in $2; do
  if [ "$v" = off ]; then unset $VAR; else export $VAR=$v; fi
  tag=$(echo $v | tr ':' '_')
  CZC_BENCH_ALLOPS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 3 > gpurun_out/env_${tag}_$r.json 2> gpurun_out/env_${tag}_$r.log
  echo "== $VAR=$v round $r: $(python -c "import json; print(json.load(open('gpurun_out/env_${tag}_$r.json'))['ms_per_step'])") ms/step"
  grep "op " gpurun_out/env_${tag}_$r.log | grep "$3" | sed 's/.*op //'
done
done
