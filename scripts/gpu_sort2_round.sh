# sort_rank2: parity tests for the sort mode + micro-benchmarks, new kernel vs the one-column-per-CU kernel
TAG=${1:-sort2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q -k "sort or ot_loop or smoke or driver" > $OUT/tests_sort.log 2>&1; echo "pytest rc=$?" >> $OUT/tests_sort.log )
for n in 16384 12544 9216 6400 4096; do
  timeout 300 python scripts/microbench.py --only sort --reps 20 --n $n --ns 12288 >> $OUT/sort_rank2.log 2>&1
  OPTEX_SORT_PATH=rank1 timeout 300 python scripts/microbench.py --only sort --reps 20 --n $n --ns 12288 >> $OUT/sort_rank1.log 2>&1
done
tail -n 5 $OUT/tests_sort.log
echo "--- rank2"; grep sort_match $OUT/sort_rank2.log | grep '"kernel": "sort_match"'
echo "--- rank1"; grep sort_match $OUT/sort_rank1.log | grep '"kernel": "sort_match"'
