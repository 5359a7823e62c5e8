#!/bin/bash
# Closing session of round 6, on the final library: the -m gpu suite and smoke(); rocprofv3 kernel-trace summaries of the bench (cdf,
# sort, chol at 64 textures per step, cdf at 8, one texture with the reference's default flags); PMC HBM traffic (FETCH_SIZE /
# WRITE_SIZE, separate passes) of the bench in cdf and sort mode, merged into profiles/pmc_traffic.json ON THE BOX so that the
# default bench line that follows cites the same library; SQ counters of the two sort-match kernels; the probes behind DESIGN's
# round-6 numbers; optex_sort_columns re-measured; the default bench line.
#     gpurun --timeout 3600 -- 'bash scripts/gpu_r06_final.sh <tag> <commit>'
TAG=${1:-r06final}
COMMIT=${2:-unknown}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STAMP="round 6, commit $COMMIT, one MI355X"
( timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log ); tail -2 $OUT/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log ); tail -2 $OUT/smoke.log
prof() {  # name, title, bench args...
  local NAME=$1 TITLE=$2; shift 2
  ( timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$NAME -o prof -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing "$@" > $OUT/prof_$NAME.log 2>&1; echo "rc=$?" >> $OUT/prof_$NAME.log )
  python scripts/summarize_rocprof.py $OUT/prof_$NAME/prof_kernel_trace.csv --warmup 1 --title "$TITLE ($STAMP)" --out $OUT/bench_${NAME}_kernel_summary.md > /dev/null 2>&1
}
for MODE in cdf sort chol; do
  prof b64_$MODE "bench.py --hist_mode $MODE, 64 textures per step" --hist_mode $MODE --other_modes ""
  rm -rf $OUT/prof_b64_$MODE
done
prof b8_cdf "bench.py --batch 8 (BASELINE config 4's per-GPU shard), cdf" --batch 8 --steps 5 --other_modes ""
python scripts/timeline_gaps.py $OUT/prof_b8_cdf/prof_kernel_trace.csv --warmup 1 --out $OUT/b8_timeline_gaps.md > /dev/null 2>&1
rm -rf $OUT/prof_b8_cdf
( timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_single -o prof -- python scripts/single_latency.py 3 > $OUT/prof_single.log 2>&1; echo "rc=$?" >> $OUT/prof_single.log )
python scripts/summarize_rocprof.py $OUT/prof_single/prof_kernel_trace.csv --all --title "ONE texture, B = 1, relu5_1..relu1_1, PCA, chol, 493 OT iterations (the reference default command line): 3 calls incl. the first ($STAMP)" --out $OUT/single_texture_kernel_summary.md > /dev/null 2>&1
rm -rf $OUT/prof_single
grep "^call" $OUT/prof_single.log
head -24 $OUT/bench_b64_sort_kernel_summary.md | cut -c1-160
for MODE in cdf sort; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $CTR -f csv -d $OUT/${MODE}_$CTR -o pmc -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/${MODE}_$CTR.log 2>&1
  done
  python scripts/summarize_pmc.py $OUT/${MODE}_FETCH_SIZE/pmc_counter_collection.csv $OUT/${MODE}_WRITE_SIZE/pmc_counter_collection.csv --out $OUT/pmc_traffic_$MODE.json --measured "$STAMP" --command "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes '' --no_kernel_timing" > $OUT/pmc_traffic_$MODE.log 2>&1
  rm -rf $OUT/${MODE}_FETCH_SIZE $OUT/${MODE}_WRITE_SIZE
done
python scripts/collect_profiles.py $TAG r06 2>&1 | tail -1      # writes profiles/pmc_traffic.json on the box: the bench below reads it
# the two sort-match kernels: instruction mix, LDS counters, wait states at [64 x 256] columns of 16384 keys
cd /tmp
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/$OUT/sortpmc_$tag -o pmc --output-format csv -- $GRAFT_REPO_ROOT/scripts/sort5_probe.bin 1 0 16384 > $GRAFT_REPO_ROOT/$OUT/sortpmc_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_sq.py $(find $OUT -name "*counter_collection.csv" -path "*sortpmc*") --match rank_match --elements 268435456 --out $OUT/sort_match5w_sq_counters.md --title "rank_match5w_kernel vs rank_match4_kernel: SQ counters, [64 x 256] columns of 16384 keys ($STAMP)" --command "rocprofv3 --kernel-trace --pmc <counters> -- scripts/sort5_probe.bin 1 0 16384" > /dev/null 2>&1
rm -rf $OUT/sortpmc_*
grep -E "^## |wave-instructions|ACTIVE_INST_VALU /" $OUT/sort_match5w_sq_counters.md
# probes
( echo "# rank_match5w_kernel against rank_match4_kernel ($STAMP): scripts/sort5_probe.bin 4 8"; scripts/sort5_probe.bin 4 8 ) 2>&1 | grep -v "amdgpu.ids" > $OUT/sort5_probe.log
grep -E "weighted" $OUT/sort5_probe.log; grep -c WRONG $OUT/sort5_probe.log
( echo "# optex_sort_columns / optex_sort_match at [64, 256, n] ($STAMP): scripts/microbench.py --only sort"; for n in 16384 9216 4096; do python scripts/microbench.py --only sort --S 64 --n $n --reps 5; done ) 2>&1 | grep -v "amdgpu.ids" > $OUT/sort_columns_microbench.log
grep sort_kv $OUT/sort_columns_microbench.log | head -4
( echo "# optex_sort_columns / optex_sort_match on rotated, quantised and half-zero columns ($STAMP): scripts/sort_ties_probe.py"; python scripts/sort_ties_probe.py 16384 9216 4096 ) 2>&1 | grep -v "amdgpu.ids" > $OUT/sort_ties_probe.log
grep "sort_columns" $OUT/sort_ties_probe.log | head -3
( echo "# optex_vgg_glue_layout, the transposing launches of a 64-texture step, alone ($STAMP): scripts/glue_planar_probe.py"; python scripts/glue_planar_probe.py ) 2>&1 | grep -v "amdgpu.ids" > $OUT/glue_planar_probe.log
( echo "# cdf_fused_kernel: shared style histogram (style range 4.5) vs style binned by every workgroup (3.0) ($STAMP): scripts/cdf_probe_ship.bin n 8 half"; for h in 4.5 3.0; do for n in 16384 12544 9216 6400 4096; do scripts/cdf_probe_ship.bin $n 8 $h; done; done ) 2>&1 | grep -v "amdgpu.ids" | grep -E "^#|shipping|behind" > $OUT/cdf_probe.log
( echo "# the two Cholesky + inverse kernels ($STAMP): scripts/chol_probe.bin"; scripts/chol_probe.bin ) 2>&1 | grep -v "amdgpu.ids" > $OUT/chol_probe.log
( timeout 1200 python scripts/sort_stress.py --loop 60 7 ) 2>&1 | grep -v "amdgpu.ids" > $OUT/sort_stress_loop.log; tail -1 $OUT/sort_stress_loop.log
( timeout 1800 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 600 $OUT/bench_default.json; echo
tail -2 $OUT/bench_default.err
