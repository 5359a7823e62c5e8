# kernel-trace of a short bench run in another hist_mode:  bash scripts/gpu_prof_mode.sh chol
MODE=${1:-chol}
OUT=gpurun_out/prof_$MODE
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" > $OUT/bench.log 2>&1
tail -n 1 $OUT/bench.log | cut -c1-400
python scripts/summarize_rocprof.py $OUT/prof/prof_kernel_trace.csv --warmup 1 --out $OUT/summary.md | head -40
