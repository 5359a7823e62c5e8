# new rank-based sort: parity tests + micro-benchmarks; GEMM variants with long runs (stable clocks)
mkdir -p gpurun_out/sort2
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "sort" > gpurun_out/sort2/tests_rank.log 2>&1
OPTEX_SORT_PATH=radix timeout 900 python -m pytest tests -m gpu -x -q -k "sort" > gpurun_out/sort2/tests_radix.log 2>&1
timeout 300 python scripts/microbench.py --only sort --reps 20 > gpurun_out/sort2/sort_rank.log 2>&1
timeout 300 python scripts/microbench.py --only sort --reps 20 --n 4096 --ns 4096 >> gpurun_out/sort2/sort_rank.log 2>&1
OPTEX_SORT_PATH=radix timeout 300 python scripts/microbench.py --only sort --reps 5 > gpurun_out/sort2/sort_radix.log 2>&1
for cfg in 0 2 10 12 24; do
  echo "== cfg $cfg" >> gpurun_out/sort2/gemm_long.log
  OPTEX_GEMM_CFG=$cfg timeout 300 python scripts/microbench.py --only gemm --reps 600 >> gpurun_out/sort2/gemm_long.log 2>&1
done
tail -3 gpurun_out/sort2/tests_rank.log gpurun_out/sort2/tests_radix.log
cat gpurun_out/sort2/sort_rank.log gpurun_out/sort2/sort_radix.log
grep -E "==|_rotate" gpurun_out/sort2/gemm_long.log
