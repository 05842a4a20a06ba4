#!/bin/bash
# dev: board power and clocks sampled beside a bench run (rocm-smi reads sysfs; no settings are changed)
mkdir -p gpurun_out
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --kernel-table 0 > gpurun_out/pw_bench.json 2> gpurun_out/pw_bench.log &
BP=$!
for i in $(seq 1 70); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Package Power\|sclk" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 0.3
done > gpurun_out/pw_samples.txt
wait $BP
echo idle >> gpurun_out/pw_samples.txt
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | tr '\n' ';' >> gpurun_out/pw_samples.txt
python -c "
import json; d=json.load(open('gpurun_out/pw_bench.json')); print('ms_per_step', d['ms_per_step'])"
cat gpurun_out/pw_samples.txt
