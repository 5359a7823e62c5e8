# round 2, GPU call Z: 4096 keys on 512 x 8; sort tests; bench in sort mode
OUT=gpurun_out/${1:-r02z}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=|Error|error" $OUT/pytest_sort.log | tail -n 12
for E in 2 1; do
  OPTEX_SORT_EXTRA_NT=$E timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n 4096 --ns 4096 --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 extra_nt=$E n=4096 /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-230
( timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --hist_mode sort --other_modes "" > $OUT/bench_sort.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
python -c "
import json; d=json.loads(open('$OUT/bench_sort.json').read().strip().splitlines()[-1]); print(d['value']); print([ (k['kernel'],k['frac'],k['avg_us'],k['launches']) for k in d.get('kernels',[])])"
tail -3 $OUT/bench.err
