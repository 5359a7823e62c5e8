#!/bin/bash
# Round 5, session E: sort variants side by side (round-4 kernel, refactored one-column kernel, persistent at 128 registers, 512 x 32 keys
# at 128 registers); GEMM in-loop loads in two halves vs one
OUT=gpurun_out/r05e
mkdir -p $OUT
export TMPDIR=/tmp
( echo "## round-4 kernel (commit f16fea3), same box"; scripts/sort_time_probe_r04.bin 4; echo "## current sources"; scripts/sort_time_probe.bin 4 ) > $OUT/sort_time_probe.log 2>&1
cat $OUT/sort_time_probe.log
( for B in gemm_rs_probe_d16.bin gemm_rs_probe_nosplit.bin; do echo "## $B"; for D in 0 2 1; do scripts/$B 64 16384 256 256 20 0 $D; done; scripts/$B 8 4096 256 256 50 0 0; scripts/$B 8 9216 256 256 50 0 0; scripts/$B 64 16384 128 128 20 0 0; done ) > $OUT/gemm_probe.log 2>&1
cat $OUT/gemm_probe.log
P=scripts/gemm_timeline_probe.bin
( $P $OUT/tl_rs_b64.bin 0 64 16384 0 0; $P $OUT/tl_rs_b64_rowstat.bin 0 64 16384 1 0 ) > $OUT/timeline_probe.log 2>&1
python scripts/gemm_timeline_report.py $OUT/tl_rs_b64.bin $OUT/tl_rs_b64_rowstat.bin > $OUT/gemm_timeline.md 2> $OUT/report.err
rm -f $OUT/tl_*.bin
grep "^## \|entry ->\|issued ->\|whole tile\|epilogue\|effective\|k-steps" $OUT/gemm_timeline.md | head -30 | cut -c1-200
( timeout 600 python -m pytest tests/test_gpu_gemm_rs.py tests/test_gpu_parity.py -m gpu -q -k "gemm or ot_loop" 2>&1 | tail -3 ) > $OUT/pytest_subset.log 2>&1
cat $OUT/pytest_subset.log
