# round 2, GPU call N: Gram kernel with batched asynchronous loads: linear-mode tests, microbench, bench by mode
OUT=gpurun_out/${1:-r02n}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_linalg.py -m gpu -q -k "linear or chol or pca or sym or chain or cov or stats or epilogue" > $OUT/pytest_sel.log 2>&1; echo "rc=$?" >> $OUT/pytest_sel.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_sel.log | tail -n 8
for GK in 16 32; do
  OPTEX_GRAM_GK=$GK timeout 300 python scripts/microbench.py --only linear --S 64 2>/dev/null | grep -E "gram|cov_fin" | sed "s/^/gk=$GK /"
done | tee $OUT/microbench_gram.log | cut -c1-200
( timeout 600 python scripts/microbench.py --only linalg --S 64 > $OUT/microbench.log 2>&1; echo "rc=$?" >> $OUT/microbench.log )
grep -E "loop_chol\"" $OUT/microbench.log | cut -c1-200
( timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --hist_mode chol --other_modes pca,sym,fused,refdefaults > $OUT/bench_chol.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
python -c "
import json; d=json.loads(open('$OUT/bench_chol.json').read().strip().splitlines()[-1]); print(d['value'], d.get('textures_per_s_by_hist_mode'), d.get('textures_per_s_fused_by_hist_mode'), d.get('textures_per_s_reference_defaults')); print([ (k['kernel'],k['frac'],k['avg_us'],k['launches']) for k in d.get('kernels',[])])"
tail -3 $OUT/bench.err
