#!/bin/bash
# Round 5, session F: GEMM variants side by side (double-buffered accumulators / ring depth / flush positions / two waves per SIMD);
# sort with 8 instead of 4 keys' LDS operations in flight; parity of the GEMM paths
OUT=gpurun_out/r05f
mkdir -p $OUT
export TMPDIR=/tmp
( for B in d16 nodb nodb8 db16 flush4 wpe2 wpe2d8; do echo "## gemm_rs_probe_$B.bin"; scripts/gemm_rs_probe_$B.bin 64 16384 256 256 20 0 0; scripts/gemm_rs_probe_$B.bin 64 16384 256 256 20 0 1; scripts/gemm_rs_probe_$B.bin 64 4096 256 256 20 0 0; for N in 4096 9216 16384; do scripts/gemm_rs_probe_$B.bin 8 $N 256 256 50 0 0; done; done ) > $OUT/gemm_probe.log 2>&1
cat $OUT/gemm_probe.log
P=scripts/gemm_timeline_probe.bin
( $P $OUT/tl_rs_b64.bin 0 64 16384 0 0; $P $OUT/tl_rs_b64_rowstat.bin 0 64 16384 1 0; $P $OUT/tl_rs_b8_4096.bin 0 8 4096 1 0 ) > $OUT/timeline_probe.log 2>&1
cat $OUT/timeline_probe.log
python scripts/gemm_timeline_report.py $OUT/tl_rs_b64.bin $OUT/tl_rs_b64_rowstat.bin $OUT/tl_rs_b8_4096.bin > $OUT/gemm_timeline.md 2> $OUT/report.err
rm -f $OUT/tl_*.bin
grep "^## \|entry ->\|issued ->\|whole tile\|epilogue\|effective" $OUT/gemm_timeline.md | cut -c1-200
( echo "## R4_G4 = 4 (shipping)"; scripts/sort_time_probe.bin 4; echo "## R4_G4 = 8"; scripts/sort_time_probe_g8.bin 4 ) > $OUT/sort_g4_g8.log 2>&1
grep "schedule-weighted\|^##" $OUT/sort_g4_g8.log
( timeout 900 python -m pytest tests/test_gpu_gemm_rs.py tests/test_gpu_parity.py tests/test_gpu_linalg.py -m gpu -q -k "gemm or ot_loop or linear or collapsed or chain" 2>&1 | tail -3 ) > $OUT/pytest_subset.log 2>&1
cat $OUT/pytest_subset.log
