# round 2, GPU call D: scaled true-product Newton-Schulz (linalg tests, forward fixtures), linalg microbench, bench by mode
OUT=gpurun_out/r02d
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_linalg.py -m gpu -q -s --durations=5 > $OUT/pytest_linalg.log 2>&1; echo "rc=$?" >> $OUT/pytest_linalg.log )
( timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "forward_matches" > $OUT/pytest_forward.log 2>&1; echo "rc=$?" >> $OUT/pytest_forward.log )
( timeout 600 python scripts/microbench.py --only linalg --S 64 > $OUT/microbench_linalg.log 2>&1; echo "rc=$?" >> $OUT/microbench_linalg.log )
( timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_linalg.log | tail -n 20
grep -E "passed|failed|FAILED|max err|rc=|PCA ranks" $OUT/pytest_forward.log | tail -n 20
grep -E "spd_sqrt|transfer_|loop_pca\"|loop_sym\"" $OUT/microbench_linalg.log | cut -c1-200
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d.get('textures_per_s_by_hist_mode'), d.get('textures_per_s_fused_rotations'))"
