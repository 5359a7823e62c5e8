mkdir -p gpurun_out/sort3
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "sort" > gpurun_out/sort3/tests_rank.log 2>&1
timeout 300 python scripts/microbench.py --only sort --reps 20 > gpurun_out/sort3/sort_rank.log 2>&1
timeout 300 python scripts/microbench.py --only sort --reps 20 --n 4096 --ns 4096 >> gpurun_out/sort3/sort_rank.log 2>&1
tail -n 3 gpurun_out/sort3/tests_rank.log
cat gpurun_out/sort3/sort_rank.log
