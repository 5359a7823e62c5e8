# GEMM structure experiments on the headline shape (scratch results under gpurun_out/)
mkdir -p gpurun_out/gemm2
export TMPDIR=/tmp
for cfg in 0 10 11 12 13 14 20 21 22 23 24 25; do
  echo "== cfg $cfg" >> gpurun_out/gemm2/variants.log
  OPTEX_GEMM_CFG=$cfg timeout 300 python scripts/microbench.py --only gemm --reps 20 >> gpurun_out/gemm2/variants.log 2>&1
  OPTEX_GEMM_CFG=$cfg timeout 300 python scripts/microbench.py --only gemm --reps 20 --n 4096 >> gpurun_out/gemm2/variants.log 2>&1
done
for cfg in 10 12 20 22; do
  echo "== tests cfg $cfg" >> gpurun_out/gemm2/tests.log
  OPTEX_GEMM_CFG=$cfg timeout 600 python -m pytest tests -m gpu -x -q -k "gemm or ot_loop or full_size or optimal_transport" >> gpurun_out/gemm2/tests.log 2>&1
done
for cfg in 10 20; do
timeout 300 env OPTEX_GEMM_CFG=$cfg rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -f csv -d gpurun_out/gemm2/pmc_sq_$cfg -o pmc -- python scripts/microbench.py --only gemm --reps 3 > gpurun_out/gemm2/pmc_sq_$cfg.log 2>&1
done
grep -E "==|rotate" gpurun_out/gemm2/variants.log
grep -E "==|passed|failed|error" gpurun_out/gemm2/tests.log
