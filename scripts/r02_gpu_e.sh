# round 2, GPU call E: owner-ranked sort match kernel (sort_rank3.hip): parity tests, microbench against the slot-ranked
# kernel (OPTEX_SORT_PATH=rank2) at the five pass sizes, SQ counter passes for the new kernel
OUT=gpurun_out/r02e
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_sort.log | tail -n 20
for N in 16384 12544 9216 6400 4096; do
  NS=$((N*3/4))
  for P in rank3 rank2; do
    OPTEX_SORT_PATH=$P timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $NS --reps 10 2>/dev/null | grep sort_match | sed "s/^/$P n=$N /"
  done
done | tee $OUT/microbench_sortmatch.log | cut -c1-200
MB="python scripts/microbench.py --only sortmatch --S 64 --reps 6"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $OUT/sort_sq1 -o pmc -- $MB > $OUT/sort_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -f csv -d $OUT/sort_sq2 -o pmc -- $MB > $OUT/sort_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN -f csv -d $OUT/sort_sq3 -o pmc -- $MB > $OUT/sort_sq3.log 2>&1
python scripts/summarize_sq.py $OUT/sort_sq1/pmc_counter_collection.csv $OUT/sort_sq2/pmc_counter_collection.csv $OUT/sort_sq3/pmc_counter_collection.csv --match rank_match3 --skip 3 --elements $((64*256*16384)) --title "rank_match3_kernel ([64, 256, 16384] against a [1, 256, 12288] style): instruction mix and wait states" --command "rocprofv3 --kernel-trace --pmc <counters> -- $MB" --out $OUT/sort_match3_sq_counters.md > /dev/null 2>&1
rm -rf $OUT/sort_sq1 $OUT/sort_sq2 $OUT/sort_sq3
tail -n 32 $OUT/sort_match3_sq_counters.md
