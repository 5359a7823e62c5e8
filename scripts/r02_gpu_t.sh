# round 2, GPU call T: counting-idiom throughput, per-phase wall clock of rank_match3 / rank_match4
OUT=gpurun_out/${1:-r02t}
mkdir -p $OUT
timeout 120 scripts/valu_lds_rate_probe.bin 2>&1 | grep -E "k_count" | tee $OUT/probe_count.log
for R in 3 4; do timeout 120 scripts/sort_rank${R}_probe.bin 16384 12288 2>&1 | tee $OUT/phases_rank${R}.log; done
timeout 120 scripts/sort_rank4_probe.bin 9216 6912 2>&1 | tee $OUT/phases_rank4_9216.log
timeout 120 scripts/sort_rank3_probe.bin 9216 6912 2>&1 | tee $OUT/phases_rank3_9216.log
