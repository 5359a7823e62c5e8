# round 2: rocprofv3 kernel-trace summary of the bench in sort mode (final library)
OUT=gpurun_out/${1:-r02profsort}
mkdir -p $OUT
export TMPDIR=/tmp
MODE=sort
( timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$MODE -o prof -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --hist_mode $MODE --other_modes "" > $OUT/prof_bench_$MODE.log 2>&1; echo "prof rc=$?" >> $OUT/prof_bench_$MODE.log )
python scripts/summarize_rocprof.py $OUT/prof_$MODE/prof_kernel_trace.csv --warmup 1 --out $OUT/summary_$MODE.md > /dev/null 2>&1
grep -E "rank_match4|rank_columns|sort_columns|kernel busy" $OUT/summary_$MODE.md | cut -c1-160
rm -rf $OUT/prof_$MODE
