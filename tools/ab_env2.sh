#!/bin/bash
# dev: step time under two environment settings, interleaved on ONE device inside one gpurun call:
#   bash tools/ab_env2.sh "CZC_EMA_FOLD=0" "CZC_EMA_FOLD=1" [rounds] [steps]
set -e
mkdir -p gpurun_out
R=${3:-3}
S=${4:-40}
for r in $(seq 1 $R); do
  i=0
  for v in "$1" "$2"; do
    i=$((i+1))
    env $v python bench.py --steps $S --warmup 8 --no-cpu-baseline --kernel-table 0 --steady-steps 0 > gpurun_out/abe_${i}_$r.json 2> gpurun_out/abe_${i}_$r.log
    python -c "
import json; d=json.load(open('gpurun_out/abe_${i}_$r.json')); print('setting [$v] round $r ms_per_step', d['ms_per_step'], 'median', d['step_ms']['median'], flush=True)"
  done
done
