TAG=${1:-r1d}
bash scripts/gpu_tests_bench_prof.sh $TAG
export TMPDIR=/tmp
for n in 16384 12544 9216 6400 4096; do
  timeout 300 python scripts/microbench.py --only sort --reps 20 --n $n --ns 12288 >> gpurun_out/$TAG/sort_micro.log 2>&1
done
grep '"kernel": "sort_match"\|"kernel": "sort_columns"' gpurun_out/$TAG/sort_micro.log | grep -v '_sort_match", "kernel": "sort_columns"'
