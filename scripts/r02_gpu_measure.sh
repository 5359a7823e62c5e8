# round 2, measurement call: every fraction quoted in DESIGN / README must be reproducible from a file under profiles/.
#   1. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of bench.py in cdf AND sort mode -> pmc_traffic json
#   2. SQ instruction / wait counters of the sort match kernel and of the rotation GEMM (MFMA utilisation) on the microbench
#   3. socket power + shader clock (rocm-smi) while the rotation GEMM runs back to back
# Usage: bash scripts/r02_gpu_measure.sh <tag>
TAG=${1:-r02m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 1 --warmup 1 --no_cpu_baseline --other_modes \"\" --no_kernel_timing"
for MODE in cdf sort; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 420 rocprofv3 --kernel-trace --pmc $CTR -f csv -d $OUT/${MODE}_$CTR -o pmc -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/${MODE}_$CTR.log 2>&1
  done
  python scripts/summarize_pmc.py $OUT/${MODE}_FETCH_SIZE/pmc_counter_collection.csv $OUT/${MODE}_WRITE_SIZE/pmc_counter_collection.csv --out $OUT/pmc_traffic_$MODE.json --command "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --hist_mode $MODE --steps 1 --warmup 1 --no_cpu_baseline --other_modes '' --no_kernel_timing" > $OUT/pmc_traffic_$MODE.log 2>&1
  rm -rf $OUT/${MODE}_FETCH_SIZE $OUT/${MODE}_WRITE_SIZE
done
# sort match kernel: instruction mix and wait states (two SQ passes of <= 8 counters each)
MB="python scripts/microbench.py --only sortmatch --S 64 --reps 6"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $OUT/sort_sq1 -o pmc -- $MB > $OUT/sort_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -f csv -d $OUT/sort_sq2 -o pmc -- $MB > $OUT/sort_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN -f csv -d $OUT/sort_sq3 -o pmc -- $MB > $OUT/sort_sq3.log 2>&1
python scripts/summarize_sq.py $OUT/sort_sq1/pmc_counter_collection.csv $OUT/sort_sq2/pmc_counter_collection.csv $OUT/sort_sq3/pmc_counter_collection.csv --match rank_match --skip 3 --elements $((64*256*16384)) --title "rank_match_kernel ([64, 256, 16384] against a [1, 256, 12288] style): instruction mix and wait states" --command "rocprofv3 --kernel-trace --pmc <counters> -- $MB" --out $OUT/sort_match_sq_counters.md > /dev/null 2>&1
# rotation GEMM: MFMA utilisation
MG="python scripts/microbench.py --only gemm --S 64 --reps 30"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU GRBM_GUI_ACTIVE -f csv -d $OUT/gemm_sq -o pmc -- $MG > $OUT/gemm_sq.log 2>&1
python scripts/summarize_sq.py $OUT/gemm_sq/pmc_counter_collection.csv --match gemm16_cm --skip 3 --title "gemm16_cm_kernel (M = K = 256, n = 64 x 16384): MFMA utilisation" --command "rocprofv3 --kernel-trace --pmc <counters> -- $MG" --out $OUT/gemm_mfma_counters.md > /dev/null 2>&1
rm -rf $OUT/sort_sq1 $OUT/sort_sq2 $OUT/sort_sq3 $OUT/gemm_sq
# power / clock while the GEMM runs back to back (un-profiled)
rocm-smi --showpower --showclocks --showmaxpower > $OUT/power_idle.txt 2>&1
( for i in $(seq 1 150); do echo "t=$(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; sleep 0.2; done ) > $OUT/power_samples.txt 2>&1 &
SAMPLER=$!
sleep 2
echo "start_gemm=$(date +%s.%N)" > $OUT/power_marks.txt
timeout 120 python scripts/microbench.py --only gemm --S 64 --reps 2500 > $OUT/power_gemm_long.log 2>&1
echo "end_gemm=$(date +%s.%N)" >> $OUT/power_marks.txt
sleep 1
kill $SAMPLER 2>/dev/null
cat $OUT/sort_match_sq_counters.md | tail -n 30
cat $OUT/gemm_mfma_counters.md | tail -n 20
grep -c Power $OUT/power_samples.txt; cat $OUT/power_gemm_long.log | head -3
python - <<PY
import json
for m in ("cdf", "sort"):
    try:
        d = json.load(open("$OUT/pmc_traffic_%s.json" % m))
        print(m, {k: round(v["hbm_bytes"] / 1e6, 1) for k, v in d["kernels"].items()})
    except Exception as e:
        print(m, "failed", e)
PY
