# round 2, final record: whole -m gpu suite, smoke, default bench (-> profiles/r02_bench_default.json), rocprofv3 kernel trace
# of the bench (cdf and chol), PMC HBM traffic of the bench in cdf and sort mode (-> profiles/pmc_traffic.json), SQ counters
# of the sort match kernel.  Everything lands under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
TAG=${1:-r02final}
COMMIT=${2:-unknown}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_gpu.log | tail -n 6
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log ); tail -2 $OUT/smoke.log
( timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 600 $OUT/bench_default.json; echo
# kernel trace (timed steps only are summarised)
for MODE in cdf chol; do
  ( timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$MODE -o prof -- python bench.py --hist_mode $MODE --steps 2 --warmup 1 --no_cpu_baseline --other_modes "" > $OUT/prof_$MODE.log 2>&1; echo "rc=$?" >> $OUT/prof_$MODE.log )
  python scripts/summarize_rocprof.py $OUT/prof_$MODE/prof_kernel_trace.csv --warmup 1 --out $OUT/bench_b64_${MODE}_kernel_summary.md > /dev/null 2>&1
  rm -rf $OUT/prof_$MODE
done
head -30 $OUT/bench_b64_cdf_kernel_summary.md
# PMC traffic
for MODE in cdf sort; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 420 rocprofv3 --kernel-trace --pmc $CTR -f csv -d $OUT/${MODE}_$CTR -o pmc -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/${MODE}_$CTR.log 2>&1
  done
  python scripts/summarize_pmc.py $OUT/${MODE}_FETCH_SIZE/pmc_counter_collection.csv $OUT/${MODE}_WRITE_SIZE/pmc_counter_collection.csv --out $OUT/pmc_traffic_$MODE.json --measured "round 2, commit $COMMIT, one MI355X" --command "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes '' --no_kernel_timing" > $OUT/pmc_traffic_$MODE.log 2>&1
  rm -rf $OUT/${MODE}_FETCH_SIZE $OUT/${MODE}_WRITE_SIZE
done
python - <<PY
import json
for m in ("cdf", "sort"):
    try:
        d = json.load(open("$OUT/pmc_traffic_%s.json" % m))
        print(m, {k: round(v["hbm_bytes"] / 1e6, 1) for k, v in d["kernels"].items()})
    except Exception as e:
        print(m, "failed", e)
PY
# sort match kernel: instruction mix and wait states
MB="python scripts/microbench.py --only sortmatch --S 64 --reps 6"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $OUT/sort_sq1 -o pmc -- $MB > $OUT/sort_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -f csv -d $OUT/sort_sq2 -o pmc -- $MB > $OUT/sort_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN -f csv -d $OUT/sort_sq3 -o pmc -- $MB > $OUT/sort_sq3.log 2>&1
python scripts/summarize_sq.py $OUT/sort_sq1/pmc_counter_collection.csv $OUT/sort_sq2/pmc_counter_collection.csv $OUT/sort_sq3/pmc_counter_collection.csv --match rank_match3 --skip 3 --elements $((64*256*16384)) --title "rank_match3_kernel ([64, 256, 16384] against a [1, 256, 12288] style): instruction mix and wait states" --command "rocprofv3 --kernel-trace --pmc <counters> -- $MB" --out $OUT/sort_match3_sq_counters.md > /dev/null 2>&1
rm -rf $OUT/sort_sq1 $OUT/sort_sq2 $OUT/sort_sq3
tail -n 12 $OUT/sort_match3_sq_counters.md
