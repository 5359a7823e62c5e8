#!/bin/bash
# Round 5, session J: the persistent double-buffered fused cdf matcher — parity tests, phase knock-outs, loop micro-benchmark
OUT=gpurun_out/r05j
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x -k "cdf or ot_loop or hist_match or optimal_transport" 2>&1 | tail -8 ) > $OUT/pytest_cdf.log 2>&1
tail -4 $OUT/pytest_cdf.log
for n in 16384 12544 9216 6400 4096; do
  for v in ship; do
    timeout 120 scripts/cdf_probe_$v.bin $n 8 2>&1 | grep -v amdgpu.ids >> $OUT/cdf_probe.log
  done
done
for v in noatomic nolut noapply nohist_noapply; do
  timeout 120 scripts/cdf_probe_$v.bin 16384 8 2>&1 | grep -v "amdgpu.ids\|copy\|checksum" >> $OUT/cdf_probe.log
done
cat $OUT/cdf_probe.log
for n in 16384 4096; do
  ( timeout 300 python scripts/microbench.py --S 64 --n $n --ns $((n * 3 / 4)) --only loop ) 2>&1 | grep -v "amdgpu.ids\|gemm_tn\|col_minmax" | cut -c1-200
  ( timeout 300 python scripts/microbench.py --S 8 --n $n --ns $((n * 3 / 4)) --only loop ) 2>&1 | grep -v "amdgpu.ids\|gemm_tn\|col_minmax" | cut -c1-200
done
