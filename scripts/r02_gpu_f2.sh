# round 2, GPU call F2: per-phase wall clock with ONE workgroup per CU (LDS padded) against two
OUT=gpurun_out/${1:-r02f2}
mkdir -p $OUT
for PAD in 0 4096; do echo "LDS pad $PAD"; OPTEX_SORT_LDS_PAD=$PAD timeout 120 scripts/sort_rank4_probe.bin 16384 12288 2>&1; done | tee $OUT/phases_rank4_one_vs_two.log
for PAD in 0 60000; do echo "LDS pad $PAD"; OPTEX_SORT_LDS_PAD=$PAD timeout 120 scripts/sort_rank4_probe.bin 9216 9216 2>&1; done | tee -a $OUT/phases_rank4_one_vs_two.log
