# round 2, GPU call U: rank_match4 after the register-pressure fixes: parity, microbench, phases
OUT=gpurun_out/${1:-r02u}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=|Error|error" $OUT/pytest_sort.log | tail -n 12
for N in 16384 12544 9216 6400 4096; do
  NS=$((N*3/4))
  timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $NS --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 n=$N ns=$NS /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-220
timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n 16384 --ns 16384 --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 n=16384 ns=16384 /" | tee -a $OUT/microbench_sortmatch.log | cut -c1-220
for A in "16384 12288" "12544 9408" "9216 6912"; do timeout 120 scripts/sort_rank4_probe.bin $A 2>&1 | tee -a $OUT/phases_rank4.log; done
