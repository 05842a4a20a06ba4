#!/bin/bash
# Runs the GPU kernel tests group by group (one pytest process per group, so a fault in one
# group does not hide the others).  Stops at the first group that TIMES OUT or is killed.
mkdir -p gpurun_out
groups=("ema or adamw or grad_norm or enqueue or dropout_mask or residual_ln or embed_ln or pool or colsum" "gemm" "attention" "infonce")
names=(rows gemm attn head)
rc_all=0
for i in "${!groups[@]}"; do
  echo "=== group ${names[$i]}" | tee -a gpurun_out/unit.log
  timeout -k 10 420 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "${groups[$i]}" -p no:cacheprovider > gpurun_out/unit_${names[$i]}.log 2>&1
  rc=$?
  tail -n 30 gpurun_out/unit_${names[$i]}.log | tee -a gpurun_out/unit.log
  echo "=== group ${names[$i]} rc=$rc" | tee -a gpurun_out/unit.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL: stopping" | tee -a gpurun_out/unit.log; exit $rc; fi
  if [ $rc -ne 0 ]; then rc_all=$rc; fi
done
exit $rc_all
