# round 2, GPU call P2: optex_sort_columns on rank_match4_kernel<..., SORT_EMIT>: parity, microbench (keys + indices)
OUT=gpurun_out/${1:-r02p2}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=|Error|error" $OUT/pytest_sort.log | tail -n 12; tail -4 $OUT/pytest_sort.log
for P in "" rank1; do
for N in 16384 12544 9216 4096; do
  OPTEX_SORT_PATH=$P timeout 300 python scripts/microbench.py --only sort --S 64 --n $N --reps 10 2>/dev/null | grep '"kernel": "sort' | sed "s/^/emit path=${P:-rank4} n=$N /"
done; done | tee $OUT/microbench_sort.log | cut -c1-230
