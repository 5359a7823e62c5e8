# round 2, GPU call E2: 6-slot windows where buckets are sparse, two windows in flight up to 10 keys per thread
OUT=gpurun_out/${1:-r02e2}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=|Error|error" $OUT/pytest_sort.log | tail -n 12
for N in 16384 12544 9216 6400 4096 5120 8192; do
  timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $N --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 n=$N ns=$N /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-230
