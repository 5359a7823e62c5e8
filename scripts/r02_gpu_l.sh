# round 2, GPU call L: GEMM-epilogue statistics (new tests), sort prefetch on/off, linalg microbench, bench (all modes)
OUT=gpurun_out/${1:-r02l}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_linalg.py -m gpu -q -k "epilogue or ot_loop or sort or linear or chain" > $OUT/pytest_sel.log 2>&1; echo "rc=$?" >> $OUT/pytest_sel.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_sel.log | tail -n 8
for PF in -1 0 256 1024; do
  for N in 16384 9216; do
    NS=$((N*3/4))
    OPTEX_SORT_PREFETCH=$PF timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $NS --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/prefetch=$PF n=$N /"
  done
done | tee $OUT/microbench_sort_prefetch.log | cut -c1-200
( timeout 600 python scripts/microbench.py --only linalg,linear,cdf,loop --S 64 > $OUT/microbench.log 2>&1; echo "rc=$?" >> $OUT/microbench.log )
grep -E "loop_chol\"|_linear\"|loop_cdf\"|_cdf\"" $OUT/microbench.log | cut -c1-200
( timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d.get('textures_per_s_by_hist_mode'), d.get('textures_per_s_fused_by_hist_mode'), d.get('textures_per_s_reference_defaults')); print([ (k['kernel'],k['frac'],k['avg_us'],k['launches']) for k in d.get('kernels',[])]); print([ (k['kernel'],k['frac'],k['avg_us']) for k in d.get('sort_kernels',[])])"
tail -3 $OUT/bench.err
