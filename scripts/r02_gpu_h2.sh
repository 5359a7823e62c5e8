OUT=gpurun_out/${1:-r02h2}
mkdir -p $OUT
for V in 10 13 16; do for A in "16384 16384" "12544 12544" "11264 11264"; do echo "pairmax $V: $A"; timeout 120 scripts/sort_rank4_probe_p$V.bin $A 2>&1 | grep -E "kernel|rank|queue"; done; done | tee $OUT/phases_rank4_pairmax.log
