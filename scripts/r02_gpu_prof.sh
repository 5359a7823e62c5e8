# round 2: rocprofv3 kernel-trace summaries of the bench (headline cdf mode and sort mode), PMC HBM traffic of the sort mode
OUT=gpurun_out/${1:-r02prof}
mkdir -p $OUT
export TMPDIR=/tmp
for MODE in cdf sort; do
( timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$MODE -o prof -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --hist_mode $MODE --other_modes "" > $OUT/prof_bench_$MODE.log 2>&1; echo "prof rc=$?" >> $OUT/prof_bench_$MODE.log )
python scripts/summarize_rocprof.py $OUT/prof_$MODE/prof_kernel_trace.csv --warmup 1 --out $OUT/summary_$MODE.md > /dev/null 2>&1
head -45 $OUT/summary_$MODE.md
rm -rf $OUT/prof_$MODE
done
