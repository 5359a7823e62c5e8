# round 2, GPU call M: Gram chunk depth 16 vs 32, parallel from-parts reductions, sort without the prefetch, bench
OUT=gpurun_out/${1:-r02m}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_linalg.py -m gpu -q > $OUT/pytest_sel.log 2>&1; echo "rc=$?" >> $OUT/pytest_sel.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_sel.log | tail -n 8
for GK in 16 32; do
  OPTEX_GRAM_GK=$GK timeout 300 python scripts/microbench.py --only linear --S 64 2>/dev/null | grep -E "gram|cov_fin|col_mean" | sed "s/^/gk=$GK /"
done | tee $OUT/microbench_gram.log | cut -c1-200
( timeout 600 python scripts/microbench.py --only linalg,loop,sortmatch --S 64 > $OUT/microbench.log 2>&1; echo "rc=$?" >> $OUT/microbench.log )
grep -E "loop_chol\"|loop_cdf\"|_sort_match\"" $OUT/microbench.log | cut -c1-200
( timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d.get('textures_per_s_by_hist_mode'), d.get('textures_per_s_fused_by_hist_mode'), d.get('textures_per_s_reference_defaults')); print([ (k['kernel'],k['frac'],k['avg_us'],k['launches']) for k in d.get('kernels',[])]); print([ (k['kernel'],k['frac'],k['avg_us']) for k in d.get('sort_kernels',[])])"
tail -3 $OUT/bench.err
