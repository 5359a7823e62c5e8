#!/bin/bash
# Every measurement kept under profiles/ for one commit, in one GPU-box session:
#   gpurun --timeout 3000 -- 'bash scripts/gpu_evidence.sh <tag> <commit>'
# default bench line; rocprofv3 kernel-trace summaries of the bench (cdf, sort, chol, sym, pca-default, 8 textures per step,
# un-shared rotations, single texture); PMC HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) of the
# bench in cdf and sort mode; SQ / MFMA counters of the hot GEMMs and of the sort match kernel; probe binaries (sort phases,
# sort kernel alone, normals generator, small-batch probe, Gram shapes).  Everything lands under gpurun_out/<tag>/ with the
# commit hash inside each file; scripts/collect_profiles.py copies what is to be judged into profiles/.
TAG=${1:-evidence}
COMMIT=${2:-unknown}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STAMP="round 4, commit $COMMIT, one MI355X"
( timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 300 $OUT/bench_default.json; echo
prof() {  # name, title, bench args...
  local NAME=$1 TITLE=$2; shift 2
  ( timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$NAME -o prof -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline "$@" > $OUT/prof_$NAME.log 2>&1; echo "rc=$?" >> $OUT/prof_$NAME.log )
  python scripts/summarize_rocprof.py $OUT/prof_$NAME/prof_kernel_trace.csv --warmup 1 --title "$TITLE ($STAMP)" --out $OUT/bench_${NAME}_kernel_summary.md > /dev/null 2>&1
  rm -rf $OUT/prof_$NAME
}
for MODE in cdf sort chol sym; do
  prof b64_$MODE "bench.py --hist_mode $MODE, 64 textures per step" --hist_mode $MODE --other_modes ""
done
prof b64_pca "bench.py --hist_mode chol --pca (the reference's default flags, independent textures), 64 textures per step" --hist_mode chol --pca --other_modes ""
prof b8_cdf "bench.py --batch 8 (BASELINE config 4's per-GPU shard), cdf" --batch 8 --steps 6 --other_modes ""
( timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_ownrot -o prof -- python scripts/ownrot_step.py 3 > $OUT/prof_ownrot.log 2>&1; echo "rc=$?" >> $OUT/prof_ownrot.log )
python scripts/summarize_rocprof.py $OUT/prof_ownrot/prof_kernel_trace.csv --warmup 1 --title "un-shared rotations: 64 textures per step, one rotation sequence per texture, cdf ($STAMP)" --out $OUT/bench_b64_ownrotations_kernel_summary.md > /dev/null 2>&1
rm -rf $OUT/prof_ownrot
( timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_single -o prof -- python scripts/single_latency.py 3 > $OUT/prof_single.log 2>&1; echo "rc=$?" >> $OUT/prof_single.log )
python scripts/summarize_rocprof.py $OUT/prof_single/prof_kernel_trace.csv --all --title "ONE texture, B = 1, relu5_1..relu1_1, PCA, chol, 493 OT iterations (the reference default command line): 3 calls incl. the first ($STAMP)" --out $OUT/single_texture_kernel_summary.md > /dev/null 2>&1
rm -rf $OUT/prof_single
grep "^call" $OUT/prof_single.log
head -22 $OUT/bench_b64_cdf_kernel_summary.md | cut -c1-160
for MODE in cdf sort; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $CTR -f csv -d $OUT/${MODE}_$CTR -o pmc -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/${MODE}_$CTR.log 2>&1
  done
  python scripts/summarize_pmc.py $OUT/${MODE}_FETCH_SIZE/pmc_counter_collection.csv $OUT/${MODE}_WRITE_SIZE/pmc_counter_collection.csv --out $OUT/pmc_traffic_$MODE.json --measured "$STAMP" --command "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes '' --no_kernel_timing" > $OUT/pmc_traffic_$MODE.log 2>&1
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
# hot GEMMs inside the cdf loop: rowstat variant (forward rotation) and plain variant (inverse rotation), at [64, 256, 16384]
MB="python scripts/microbench.py --only loop --S 64 --reps 3"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -f csv -d $OUT/gemm_sq -o pmc -- $MB > $OUT/gemm_sq.log 2>&1
python scripts/summarize_sq.py $OUT/gemm_sq/pmc_counter_collection.csv --match gemm --skip 2 --title "rotation GEMMs inside optex_ot_loop(cdf), [64, 256, 16384] ($STAMP)" --command "rocprofv3 --kernel-trace --pmc <SQ counters> -- $MB" --out $OUT/gemm_mfma_counters.md > /dev/null 2>&1
rm -rf $OUT/gemm_sq
grep -E "^## |MFMA util|effective" $OUT/gemm_mfma_counters.md
# sort match kernel: instruction mix, wait states, LDS
MB="python scripts/microbench.py --only loopsort --S 64"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $OUT/sort_sq1 -o pmc -- $MB > $OUT/sort_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -f csv -d $OUT/sort_sq2 -o pmc -- $MB > $OUT/sort_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN -f csv -d $OUT/sort_sq3 -o pmc -- $MB > $OUT/sort_sq3.log 2>&1
python scripts/summarize_sq.py $OUT/sort_sq1/pmc_counter_collection.csv $OUT/sort_sq2/pmc_counter_collection.csv $OUT/sort_sq3/pmc_counter_collection.csv --match rank_match4 --skip 3 --elements $((64*256*16384)) --title "rank_match4_kernel inside optex_ot_loop(sort) ([64, 256, 16384] against a [1, 256, 12288] style, range from the GEMM epilogue): instruction mix and wait states ($STAMP)" --command "rocprofv3 --kernel-trace --pmc <counters> -- $MB" --out $OUT/sort_match4_sq_counters.md > /dev/null 2>&1
rm -rf $OUT/sort_sq1 $OUT/sort_sq2 $OUT/sort_sq3
tail -n 14 $OUT/sort_match4_sq_counters.md
# probes
( echo "# rank_match4_kernel phase timestamps ($STAMP): scripts/sort_rank_probe.bin [n] [ns] [rg]"; for A in "16384 12288" "16384 12288 rg" "9216 6912 rg"; do scripts/sort_rank_probe.bin $A; done; OPTEX_SORT_LDS_PAD=4096 scripts/sort_rank_probe.bin 16384 12288 rg ) > $OUT/sort_rank4_phases.log 2>&1
( echo "# rank_match4_kernel alone, own range vs the range from the GEMM epilogue ($STAMP): scripts/sort_time_probe.bin 5"; scripts/sort_time_probe.bin 5 ) > $OUT/sort_time_probe.log 2>&1
tail -2 $OUT/sort_time_probe.log
( echo "# optex_sort_columns / optex_sort_match at [64, 256, n] ($STAMP): scripts/microbench.py --only sort"; for N in 16384 9216 4096; do python scripts/microbench.py --only sort,sortmatch --S 64 --n $N --reps 6; done ) > $OUT/sort_columns_microbench.log 2>&1
grep sort_columns $OUT/sort_columns_microbench.log | head -3
( echo "# optex_legacy_normals alone ($STAMP)"; python scripts/normals_probe.py ) > $OUT/normals_probe.log 2>&1
( echo "# bench step at 8 / 16 / 64 textures: host stream, device stream, cached rotations ($STAMP)"; timeout 900 python scripts/batch_probe.py 8 16 64 ) > $OUT/batch_probe.log 2>&1
tail -3 $OUT/batch_probe.log | cut -c1-400
( timeout 300 python scripts/gram_probe.py > $OUT/gram_probe.md 2>&1 )
( timeout 300 python scripts/ns_count_probe.py > $OUT/ns_count_probe.md 2>&1 )
