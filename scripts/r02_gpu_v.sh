# round 2, GPU call V: first-generation stagger of the two workgroups of a CU, DPP min/max
OUT=gpurun_out/${1:-r02v}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 scripts/valu_lds_rate_probe.bin 2>&1 | grep -E "HW_REG|workgroups with" | tee $OUT/probe_hwreg.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=|Error|error" $OUT/pytest_sort.log | tail -n 12
for ST in 1 0; do
for N in 16384 12544 9216; do
  NS=$((N*3/4))
  OPTEX_SORT_STAGGER=$ST timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $NS --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 stagger=$ST n=$N ns=$NS /"
done; done | tee $OUT/microbench_sortmatch.log | cut -c1-230
for A in "16384 12288"; do timeout 120 scripts/sort_rank4_probe.bin $A 2>&1 | tee -a $OUT/phases_rank4.log; done
