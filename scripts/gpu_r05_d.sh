#!/bin/bash
# Round 5, session D: persistent sort at a 128-register budget; GEMM prologue with batched loads, ring refilled a k-step late
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
( scripts/sort_time_probe.bin 4 ) > $OUT/sort_time_probe.log 2>&1
cat $OUT/sort_time_probe.log
P=scripts/gemm_timeline_probe.bin
( $P $OUT/tl_rs_b64.bin 0 64 16384 0 0
  $P $OUT/tl_rs_b64_rowstat.bin 0 64 16384 1 0
  for N in 4096 9216; do $P $OUT/tl_rs_b8_$N.bin 0 8 $N 1 0; done ) > $OUT/timeline_probe.log 2>&1
cat $OUT/timeline_probe.log
python scripts/gemm_timeline_report.py $OUT/tl_rs_b64.bin $OUT/tl_rs_b64_rowstat.bin $OUT/tl_rs_b8_4096.bin $OUT/tl_rs_b8_9216.bin > $OUT/gemm_timeline.md 2> $OUT/report.err
rm -f $OUT/tl_*.bin
( for N in 4096 6400 9216 12544 16384; do scripts/gemm_rs_probe_d16.bin 8 $N 256 256 50 0 0; scripts/gemm_rs_probe_d16.bin 8 $N 256 256 50 1 0; done
  for D in 0 2 1; do scripts/gemm_rs_probe_d16.bin 64 16384 256 256 20 0 $D; done; scripts/gemm_rs_probe_d16.bin 64 16384 256 256 20 1 0
  scripts/gemm_rs_probe_d16.bin 64 4096 256 256 20 0 0; scripts/gemm_rs_probe_d16.bin 64 4096 256 256 20 1 0
  scripts/gemm_rs_probe_d16.bin 64 16384 181 181 20 0 0; scripts/gemm_rs_probe_d16.bin 64 16384 128 128 20 0 0 ) > $OUT/gemm_probe.log 2>&1
cat $OUT/gemm_probe.log
( timeout 600 python -m pytest tests/test_gpu_gemm_rs.py tests/test_gpu_parity.py -m gpu -q -k "gemm or ot_loop or sort" 2>&1 | tail -5 ) > $OUT/pytest_subset.log 2>&1
cat $OUT/pytest_subset.log
grep -v "^| k-steps [0-9]*\.\.[0-9]* " $OUT/gemm_timeline.md | grep "^## \|entry ->\|issued ->\|whole tile\|epilogue\|effective" | cut -c1-200
grep "k-steps" $OUT/gemm_timeline.md | head -16
