#!/bin/bash
# Closing session of a round, on the final commit: the -m gpu suite, the PMC traffic passes (cdf, sort) merged into
# profiles/pmc_traffic.json ON THE BOX so that the default bench line that follows names the same commit in `traffic_source`,
# the un-shared-rotations trace, and the default bench line.   gpurun --timeout 3000 -- 'bash scripts/gpu_final.sh <tag> <commit>'
TAG=${1:-final}
COMMIT=${2:-unknown}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STAMP="round 4, commit $COMMIT, one MI355X"
( timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log ); tail -2 $OUT/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log ); tail -2 $OUT/smoke.log
for MODE in cdf sort; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $CTR -f csv -d $OUT/${MODE}_$CTR -o pmc -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/${MODE}_$CTR.log 2>&1
  done
  python scripts/summarize_pmc.py $OUT/${MODE}_FETCH_SIZE/pmc_counter_collection.csv $OUT/${MODE}_WRITE_SIZE/pmc_counter_collection.csv --out $OUT/pmc_traffic_$MODE.json --measured "$STAMP" --command "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes '' --no_kernel_timing" > $OUT/pmc_traffic_$MODE.log 2>&1
  rm -rf $OUT/${MODE}_FETCH_SIZE $OUT/${MODE}_WRITE_SIZE
done
python scripts/collect_profiles.py $TAG r04 2>&1 | tail -1      # writes profiles/pmc_traffic.json on the box: the bench below reads it
( timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_ownrot -o prof -- python scripts/ownrot_step.py 3 > $OUT/prof_ownrot.log 2>&1; echo "rc=$?" >> $OUT/prof_ownrot.log )
python scripts/summarize_rocprof.py $OUT/prof_ownrot/prof_kernel_trace.csv --warmup 1 --title "un-shared rotations: 64 textures per step, one rotation sequence per texture, cdf ($STAMP)" --out $OUT/bench_b64_ownrotations_kernel_summary.md > /dev/null 2>&1
rm -rf $OUT/prof_ownrot
grep "^step" $OUT/prof_ownrot.log
( timeout 900 python scripts/batch_probe.py 8 > $OUT/batch_probe_b8.log 2>&1 ); tail -1 $OUT/batch_probe_b8.log | cut -c1-330
( timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 400 $OUT/bench_default.json; echo
