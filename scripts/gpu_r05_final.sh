#!/bin/bash
# Closing session of round 5, on the final commit: the -m gpu suite and smoke(); rocprofv3 kernel-trace summaries of the bench (cdf,
# sort, chol, sym at 64 textures per step, cdf at 8, one texture with the reference's default flags); PMC HBM traffic (FETCH_SIZE /
# WRITE_SIZE, separate passes) of the bench in cdf and sort mode, merged into profiles/pmc_traffic.json ON THE BOX so that the
# default bench line that follows names the same commit; GEMM and sort-kernel counters; the probes behind DESIGN's numbers; the
# default bench line.      gpurun --timeout 3600 -- 'bash scripts/gpu_r05_final.sh <tag> <commit>'
TAG=${1:-r05final}
COMMIT=${2:-unknown}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STAMP="round 5, commit $COMMIT, one MI355X"
( timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log ); tail -2 $OUT/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log ); tail -2 $OUT/smoke.log
prof() {  # name, title, bench args...
  local NAME=$1 TITLE=$2; shift 2
  ( timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$NAME -o prof -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing "$@" > $OUT/prof_$NAME.log 2>&1; echo "rc=$?" >> $OUT/prof_$NAME.log )
  python scripts/summarize_rocprof.py $OUT/prof_$NAME/prof_kernel_trace.csv --warmup 1 --title "$TITLE ($STAMP)" --out $OUT/bench_${NAME}_kernel_summary.md > /dev/null 2>&1
}
for MODE in cdf sort chol sym; do
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
head -24 $OUT/bench_b64_cdf_kernel_summary.md | cut -c1-160
for MODE in cdf sort; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $CTR -f csv -d $OUT/${MODE}_$CTR -o pmc -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/${MODE}_$CTR.log 2>&1
  done
  python scripts/summarize_pmc.py $OUT/${MODE}_FETCH_SIZE/pmc_counter_collection.csv $OUT/${MODE}_WRITE_SIZE/pmc_counter_collection.csv --out $OUT/pmc_traffic_$MODE.json --measured "$STAMP" --command "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes '' --no_kernel_timing" > $OUT/pmc_traffic_$MODE.log 2>&1
  rm -rf $OUT/${MODE}_FETCH_SIZE $OUT/${MODE}_WRITE_SIZE
done
python scripts/collect_profiles.py $TAG r05 2>&1 | tail -1      # writes profiles/pmc_traffic.json on the box: the bench below reads it
# hot GEMMs inside the cdf loop at [64, 256, 16384]: matrix-pipe duty
MB="python scripts/microbench.py --only loop --S 64 --reps 3"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -f csv -d $OUT/gemm_sq -o pmc -- $MB > $OUT/gemm_sq.log 2>&1
python scripts/summarize_sq.py $OUT/gemm_sq/pmc_counter_collection.csv --match gemm --skip 2 --title "rotation GEMMs inside optex_ot_loop(cdf), [64, 256, 16384] ($STAMP)" --command "rocprofv3 --kernel-trace --pmc <SQ counters> -- $MB" --out $OUT/gemm_mfma_counters.md > /dev/null 2>&1
rm -rf $OUT/gemm_sq
grep -E "^## |MFMA util|effective" $OUT/gemm_mfma_counters.md | head -8
# the fused cdf matcher: instruction mix and LDS counters in the same loop
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $OUT/cdf_sq1 -o pmc -- $MB > $OUT/cdf_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -f csv -d $OUT/cdf_sq2 -o pmc -- $MB > $OUT/cdf_sq2.log 2>&1
python scripts/summarize_sq.py $OUT/cdf_sq1/pmc_counter_collection.csv $OUT/cdf_sq2/pmc_counter_collection.csv --match cdf_fused --skip 2 --elements $((64*256*16384)) --title "cdf_fused_kernel inside optex_ot_loop(cdf), [64, 256, 16384] ($STAMP)" --command "rocprofv3 --kernel-trace --pmc <counters> -- $MB" --out $OUT/cdf_fused_sq_counters.md > /dev/null 2>&1
rm -rf $OUT/cdf_sq1 $OUT/cdf_sq2
tail -n 12 $OUT/cdf_fused_sq_counters.md
# probes
( echo "# cdf_fused_kernel by phase ($STAMP): scripts/cdf_probe_<variant>.bin n 8"; for n in 16384 12544 9216 6400 4096; do scripts/cdf_probe_ship.bin $n 8; done; for v in ship512 noatomic nolut noapply nohist_noapply; do scripts/cdf_probe_$v.bin 16384 8; done ) 2>&1 | grep -v "amdgpu.ids" > $OUT/cdf_probe.log
( echo "# whole-column copies by shape ($STAMP): scripts/colcopy_probe.bin n"; for n in 16384 6400 4096; do scripts/colcopy_probe.bin $n; done ) 2>&1 | grep -v "amdgpu.ids" > $OUT/colcopy_probe.log
( echo "# rank_match4_kernel alone, own range vs the range from the GEMM epilogue ($STAMP): scripts/sort_time_probe.bin 5"; scripts/sort_time_probe.bin 5 ) 2>&1 | grep -v "amdgpu.ids" > $OUT/sort_time_probe.log
tail -2 $OUT/sort_time_probe.log
( echo "# bench step at 8 / 64 textures: host stream, device stream, fed device stream, cached rotations ($STAMP)"; timeout 900 python scripts/batch_probe.py 8 64 ) 2>&1 | grep -v "amdgpu.ids" > $OUT/batch_probe.log
tail -2 $OUT/batch_probe.log | tr '|' '\n' | cut -c1-160
( timeout 1800 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 600 $OUT/bench_default.json; echo
tail -2 $OUT/bench_default.err
