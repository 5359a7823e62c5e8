# round 2, GPU call Y: exact-fit workgroup sizes (576 x 16, 896 x 14, 640 x 10)
OUT=gpurun_out/${1:-r02y}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=|Error|error" $OUT/pytest_sort.log | tail -n 12
for E in 1 0; do
for N in 12544 9216 6400; do
  for NS in $((N*3/4)) $N; do
  OPTEX_SORT_EXTRA_NT=$E timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $NS --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 extra_nt=$E n=$N ns=$NS /"
done; done; done | tee $OUT/microbench_sortmatch.log | cut -c1-230
