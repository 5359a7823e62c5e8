# round 2, GPU call R: rank_match3 on 256 / 512 / 1024 threads: parity, microbench at the five pass sizes, bench sort
OUT=gpurun_out/${1:-r02r}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_sort.log | tail -n 8
for N in 16384 12544 9216 6400 4096; do
  NS=$((N*3/4))
  timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $NS --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank3 n=$N /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-200
( timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --hist_mode sort --other_modes "" > $OUT/bench_sort.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
python -c "
import json; d=json.loads(open('$OUT/bench_sort.json').read().strip().splitlines()[-1]); print(d['value']); print([ (k['kernel'],k['frac'],k['avg_us'],k['launches']) for k in d.get('kernels',[])])"
tail -3 $OUT/bench.err
