#!/bin/bash
# One GPU session that produces everything tools/make_profiles.py summarises (run from the repo root on the GPU box).
# Steps are joined with && on purpose: after a failed or killed GPU step nothing further is started.
set -o pipefail
R=$PWD
O=$R/gpurun_out/${TAG:-r3}_final
mkdir -p $O
# (below the first line per-kernel durations and counters are collected with the key encoder's second stream off:
#  kernels of two streams that share the CUs would each be charged the other's time)
python bench.py > $O/bench_default.json 2> $O/bench_default.err && tail -2 $O/bench_default.err && \
cd /tmp && export TMPDIR=/tmp && export CZC_KEY_OVERLAP=0 && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --kernel-table 0 --steady-steps 0 > $O/stats.log 2>&1 && echo stats ok && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-table 0 --steady-steps 0 > $O/pmc_fetch.log 2>&1 && echo fetch ok && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-table 0 --steady-steps 0 > $O/pmc_write.log 2>&1 && echo write ok && \
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-table 0 --steady-steps 0 > $O/pmc_sq.log 2>&1 && echo sq ok && \
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_attn -o attn -- python3 $R/tools/pmc_attn.py > $O/pmc_attn.log 2>&1 && echo attn ok
rc=$?
# raw per-dispatch csvs are large: keep the ones the summaries are made from, drop the rest
find $O -name "*agent_info.csv" -delete
du -sh $O | tail -1
exit $rc
