# round 2, GPU call B2: new shape rule: parity (all variants), microbench at the five pass sizes, 6400 both ways, bench sort
OUT=gpurun_out/${1:-r02b2}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=|Error|error" $OUT/pytest_sort.log | tail -n 12
for N in 16384 12544 9216 6400 4096; do
  timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $N --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 n=$N ns=$N /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-230
OPTEX_SORT_EXTRA_NT=0 timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n 6400 --ns 6400 --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 1024x7 n=6400 /" | tee -a $OUT/microbench_sortmatch.log | cut -c1-230
( timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --hist_mode sort --other_modes "" > $OUT/bench_sort.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
python -c "
import json; d=json.loads(open('$OUT/bench_sort.json').read().strip().splitlines()[-1]); print(d['value']); print([ (k['kernel'],k['frac'],k['avg_us'],k['launches']) for k in d.get('kernels',[])])"
tail -3 $OUT/bench.err
