# round 2, GPU call Q: transposed-MFMA epilogue of the hot-loop GEMM (16-byte stores): whole GPU suite, microbench, bench
OUT=gpurun_out/${1:-r02q}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_gpu.log | tail -n 8
( timeout 600 python scripts/microbench.py --only gemm,loop,linalg --S 64 > $OUT/microbench.log 2>&1; echo "rc=$?" >> $OUT/microbench.log )
grep -E "loop_chol\"|loop_cdf\"|rotate\"|unrotate_blend\"" $OUT/microbench.log | grep -E "gemm_tn|col_minmax|gram" | cut -c1-200
( timeout 1200 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d.get('textures_per_s_by_hist_mode'), d.get('textures_per_s_fused_by_hist_mode'), d.get('textures_per_s_reference_defaults'), d.get('textures_per_s_real_assets')); print([ (k['kernel'],k['frac'],k['avg_us'],k['launches']) for k in d.get('kernels',[])]); print([ (k['kernel'],k['frac'],k['avg_us']) for k in d.get('sort_kernels',[])])"
tail -3 $OUT/bench.err
