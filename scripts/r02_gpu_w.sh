# round 2, GPU call W: wave priorities in the short steps of rank_match4
OUT=gpurun_out/${1:-r02w}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=|Error|error" $OUT/pytest_sort.log | tail -n 12
for N in 16384 12544 9216 6400 4096; do
  NS=$((N*3/4))
  timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $NS --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 prio n=$N ns=$NS /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-230
for B in scripts/sort_rank4_probe.bin scripts/sort_rank4_probe_noprio.bin; do echo $B; timeout 120 $B 16384 12288 2>&1; done | tee -a $OUT/phases_rank4.log
