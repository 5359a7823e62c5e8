# round 2, GPU call K: the whole -m gpu suite, smoke, bench (all modes), linalg microbench with the 128-tile Gram kernel
OUT=gpurun_out/${1:-r02k}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_gpu.log | tail -n 12
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log ); tail -2 $OUT/smoke.log
( timeout 600 python scripts/microbench.py --only linalg,linear --S 64 > $OUT/microbench_linalg.log 2>&1; echo "rc=$?" >> $OUT/microbench_linalg.log )
grep -E "loop_chol\"|loop_pca\"|loop_sym\"|_linear\"" $OUT/microbench_linalg.log | cut -c1-200
( timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d.get('textures_per_s_by_hist_mode'), d.get('textures_per_s_fused_by_hist_mode'), d.get('textures_per_s_reference_defaults'), d.get('cpu_baseline',{}).get('value')); print([ (k['kernel'],k['frac'],k['avg_us']) for k in d.get('kernels',[])])"
tail -3 $OUT/bench.err
