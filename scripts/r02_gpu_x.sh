# round 2, GPU call X: SQ counters of rank_match4; two windows in flight (probe build)
OUT=gpurun_out/${1:-r02x}
mkdir -p $OUT
export TMPDIR=/tmp
for B in scripts/sort_rank4_probe.bin scripts/sort_rank4_probe_pair.bin; do echo $B; timeout 120 $B 16384 12288 2>&1; done | tee $OUT/phases_rank4_pair.log
MB="python scripts/microbench.py --only sortmatch --S 64 --reps 6"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $OUT/sort_sq1 -o pmc -- $MB > $OUT/sort_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -f csv -d $OUT/sort_sq2 -o pmc -- $MB > $OUT/sort_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN -f csv -d $OUT/sort_sq3 -o pmc -- $MB > $OUT/sort_sq3.log 2>&1
python scripts/summarize_sq.py $OUT/sort_sq1/pmc_counter_collection.csv $OUT/sort_sq2/pmc_counter_collection.csv $OUT/sort_sq3/pmc_counter_collection.csv --match rank_match4 --skip 3 --elements $((64*256*16384)) --title "rank_match4_kernel ([64, 256, 16384] against a [1, 256, 12288] style): instruction mix and wait states" --command "rocprofv3 --kernel-trace --pmc <counters> -- $MB" --out $OUT/sort_match4_sq_counters.md > $OUT/summarize.log 2>&1
rm -rf $OUT/sort_sq1 $OUT/sort_sq2 $OUT/sort_sq3
tail -n 40 $OUT/sort_match4_sq_counters.md; tail -3 $OUT/summarize.log
