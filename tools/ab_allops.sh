#!/bin/bash
# dev: per-operation table of two library variants on one device:  bash tools/ab_allops.sh "base oldsplit" 'splitk\|reduce'
set -e
mkdir -p gpurun_out
for r in 1 2; do
for v in $1; do
  if [ "$v" = base ]; then unset CZC_LIB_VARIANT; else export CZC_LIB_VARIANT=$v; fi
  CZC_BENCH_ALLOPS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 3 > gpurun_out/ops_${v}_$r.json 2> gpurun_out/ops_${v}_$r.log
  echo "== $v round $r: $(python -c "import json; print(json.load(open('gpurun_out/ops_${v}_$r.json'))['ms_per_step'])") ms/step"
  grep "op " gpurun_out/ops_${v}_$r.log | grep "$2" | sed 's/.*op //'
done
done
