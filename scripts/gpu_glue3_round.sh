TAG=${1:-glue3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 300 python -m pytest tests -m gpu -x -q -k "glue or codec or driver or config or smoke" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
( timeout 480 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --other_modes "" > $OUT/prof_bench.log 2>&1; echo "prof rc=$?" >> $OUT/prof_bench.log )
python scripts/summarize_rocprof.py $OUT/prof/prof_kernel_trace.csv --warmup 1 --out $OUT/summary.md > /dev/null 2>&1
tail -n 3 $OUT/pytest_gpu.log
grep "^{" $OUT/prof_bench.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['hot_path_ms_per_step'], r['other_ms_per_step'])"
head -24 $OUT/summary.md | cut -c1-150
