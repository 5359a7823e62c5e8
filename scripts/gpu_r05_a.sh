#!/bin/bash
# Round 5, session A: GEMM instruction timelines + small-batch GEMM times + the B = 8 baseline.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r05_a.sh'
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
P=scripts/gemm_timeline_probe.bin
( $P $OUT/tl_rs_b64.bin 0 64 16384 0 0
  $P $OUT/tl_rs_b64_rowstat.bin 0 64 16384 1 0
  $P $OUT/tl_rs_b64_zeros.bin 0 64 16384 0 1
  scripts/gemm_timeline_probe_noload.bin $OUT/tl_rs_b64_noload.bin 0 64 16384 0 0
  for N in 4096 6400 9216 12544 16384; do $P $OUT/tl_rs_b8_$N.bin 0 8 $N 0 0; done
  $P $OUT/tl_lds_b64.bin 1 64 16384 0 0
  $P $OUT/tl_lds_b64_rowstat.bin 1 64 16384 1 0
  $P $OUT/tl_lds_b8.bin 1 8 16384 0 0 ) > $OUT/timeline_probe.log 2>&1
cat $OUT/timeline_probe.log
python scripts/gemm_timeline_report.py $OUT/tl_rs_b64.bin $OUT/tl_rs_b64_rowstat.bin $OUT/tl_rs_b64_zeros.bin $OUT/tl_rs_b64_noload.bin $OUT/tl_lds_b64.bin $OUT/tl_lds_b64_rowstat.bin > $OUT/gemm_timeline_b64.md 2> $OUT/report_b64.err
python scripts/gemm_timeline_report.py $OUT/tl_rs_b8_4096.bin $OUT/tl_rs_b8_6400.bin $OUT/tl_rs_b8_9216.bin $OUT/tl_rs_b8_12544.bin $OUT/tl_rs_b8_16384.bin $OUT/tl_lds_b8.bin > $OUT/gemm_timeline_b8.md 2> $OUT/report_b8.err
tail -5 $OUT/report_b64.err $OUT/report_b8.err
# keep the raw slabs small: only the B = 8 ones travel back
rm -f $OUT/tl_rs_b64*.bin $OUT/tl_lds_b64*.bin
# un-stamped kernels at the small-batch shapes (HIP events): R-stationary vs LDS-tiled
( for N in 4096 6400 9216 12544 16384; do scripts/gemm_rs_probe_d16.bin 8 $N 256 256 50 0 0; scripts/gemm_rs_probe_d16.bin 8 $N 256 256 50 1 0; done
  scripts/gemm_rs_probe_d16.bin 64 16384 256 256 20 0 0; scripts/gemm_rs_probe_d16.bin 64 16384 256 256 20 1 0
  scripts/gemm_rs_probe_noload.bin 64 16384 256 256 20 0 0 ) > $OUT/gemm_small_batch.log 2>&1
cat $OUT/gemm_small_batch.log
( timeout 600 python scripts/batch_probe.py 8 64 ) > $OUT/batch_probe.log 2>&1
tail -3 $OUT/batch_probe.log | cut -c1-420
head -60 $OUT/gemm_timeline_b64.md
