# rocprofv3 kernel trace of a short bench run (current default settings) -> gpurun_out/<tag>/prof
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --other_modes "" > $OUT/prof_bench.log 2>&1; echo "prof rc=$?" >> $OUT/prof_bench.log )
python scripts/summarize_rocprof.py $OUT/prof/prof_kernel_trace.csv --warmup 1 --out $OUT/summary.md > /dev/null 2>&1
head -40 $OUT/summary.md
