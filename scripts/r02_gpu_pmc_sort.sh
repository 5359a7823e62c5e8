# round 2: PMC HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) of bench.py in sort mode with rank_match4
OUT=gpurun_out/${1:-r02pmcsort}
mkdir -p $OUT
export TMPDIR=/tmp
MODE=sort
for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 420 rocprofv3 --kernel-trace --pmc $CTR -f csv -d $OUT/${MODE}_$CTR -o pmc -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/${MODE}_$CTR.log 2>&1
done
python scripts/summarize_pmc.py $OUT/${MODE}_FETCH_SIZE/pmc_counter_collection.csv $OUT/${MODE}_WRITE_SIZE/pmc_counter_collection.csv --out $OUT/pmc_traffic_$MODE.json --command "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes '' --no_kernel_timing" > $OUT/pmc_traffic_$MODE.log 2>&1
tail -n 12 $OUT/pmc_traffic_$MODE.log
rm -rf $OUT/${MODE}_FETCH_SIZE $OUT/${MODE}_WRITE_SIZE
