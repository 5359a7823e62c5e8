# round 2, GPU call S: rank_match4 (float-domain, 2-cycle-VALU counting): probe additions, parity, microbench at the five pass sizes
OUT=gpurun_out/${1:-r02s}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 scripts/valu_lds_rate_probe.bin 2>&1 | grep -E "W_|dpp|L_RD128X2|L_RD64 " | tee $OUT/probe_win.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=|Error|error" $OUT/pytest_sort.log | tail -n 12
for N in 16384 12544 9216 6400 4096; do
  NS=$((N*3/4))
  timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $NS --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 n=$N ns=$NS /"
  timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $N --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 n=$N ns=$N /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-220
OPTEX_SORT_PATH=rank3 timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n 16384 --ns 12288 --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank3 n=16384 /" | tee -a $OUT/microbench_sortmatch.log | cut -c1-220
