# round 2, GPU call A: all parity tests (old + new), smoke, and a kernel trace of a chol-mode bench step (the reference's
# default hist_mode) to size the work of moving the linear modes into the C loop.
#   gpurun --timeout 2400 -- 'bash scripts/r02_gpu_a.sh'
OUT=gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log )
for MODE in chol pca; do
  ( timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$MODE -o prof -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" > $OUT/bench_$MODE.log 2>&1; echo "rc=$?" >> $OUT/bench_$MODE.log )
  python scripts/summarize_rocprof.py $OUT/prof_$MODE/prof_kernel_trace.csv --warmup 1 --out $OUT/summary_$MODE.md > /dev/null 2>&1
  rm -rf $OUT/prof_$MODE/*.db $OUT/prof_$MODE/prof_kernel_trace.csv
done
grep -E "passed|failed|error|rc=" $OUT/pytest_gpu.log | tail -n 15
tail -n 2 $OUT/smoke.log
head -n 40 $OUT/summary_chol.md
