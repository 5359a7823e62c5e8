# PMC passes over bench.py for roofline.traffic (separate runs per counter; kernel-trace only, no other trace domains)
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 1 --warmup 1 --no_cpu_baseline --other_modes \"\" --no_kernel_timing"
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o fetch -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/fetch.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o write -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/write.log 2>&1
python scripts/summarize_pmc.py $OUT/fetch/fetch_counter_collection.csv $OUT/write/write_counter_collection.csv --out $OUT/pmc_traffic.json --command "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- $CMD" > $OUT/summary.log 2>&1
tail -n 40 $OUT/summary.log
# the counter CSVs are tens of MiB each (gpurun only copies back <= 64 MiB): keep the summary only
rm -rf $OUT/fetch $OUT/write
