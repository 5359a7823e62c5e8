# round 2, GPU call B: the device linalg path (tests), PCA diagnostic, forward fixtures, bench rows per hist_mode, chol profile
OUT=gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_linalg.py tests/test_gpu_parity.py -m gpu -q -x --durations=5 > $OUT/pytest_kernels.log 2>&1; echo "rc=$?" >> $OUT/pytest_kernels.log )
( timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "forward_matches or cli" > $OUT/pytest_forward.log 2>&1; echo "rc=$?" >> $OUT/pytest_forward.log )
( timeout 600 python scripts/r02_diag_pca.py > $OUT/diag_pca.log 2>&1; echo "rc=$?" >> $OUT/diag_pca.log )
( timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
( timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_chol -o prof -- python bench.py --hist_mode chol --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" > $OUT/bench_chol.log 2>&1; echo "rc=$?" >> $OUT/bench_chol.log )
python scripts/summarize_rocprof.py $OUT/prof_chol/prof_kernel_trace.csv --warmup 1 --out $OUT/summary_chol.md > /dev/null 2>&1
rm -rf $OUT/prof_chol/*.db $OUT/prof_chol/prof_kernel_trace.csv
grep -E "passed|failed|FAILED|Error|rc=" $OUT/pytest_kernels.log | tail -n 12
grep -E "passed|failed|FAILED|max err|rc=|PCA ranks" $OUT/pytest_forward.log | tail -n 20
tail -n 40 $OUT/diag_pca.log
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r02b/bench.json").read().strip().splitlines()[-1])
    print("value", r["value"], r.get("textures_per_s_by_hist_mode"), r.get("textures_per_s_fused_rotations"))
except Exception as e:
    print("bench parse failed", e)
PY
head -n 30 $OUT/summary_chol.md | cut -c1-180
