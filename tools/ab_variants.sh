#!/bin/bash
# dev: step time of several library variants (python -m czech_contriever_amd.build --variant NAME ...), interleaved on
# ONE device inside one gpurun call:  bash tools/ab_variants.sh "base f2 f4" [rounds]
set -e
mkdir -p gpurun_out
R=${2:-2}
for r in $(seq 1 $R); do
  for v in $1; do
    if [ "$v" = base ]; then unset CZC_LIB_VARIANT; else export CZC_LIB_VARIANT=$v; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-table 0 > gpurun_out/ab_${v}_$r.json 2> gpurun_out/ab_${v}_$r.log
    python -c "
import json; d=json.load(open('gpurun_out/ab_${v}_$r.json')); print('variant $v round $r ms_per_step', d['ms_per_step'], 'ffn_up_ms', d['roofline']['kernel_ms'], flush=True)"
  done
done
