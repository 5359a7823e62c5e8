#!/bin/bash
# Round 5, session I: phase knock-outs of the fused cdf matcher
OUT=gpurun_out/r05i
mkdir -p $OUT
for n in 16384 4096; do
  for v in ship noatomic nohist nolut noapply nohist_noapply; do
    timeout 120 scripts/cdf_probe_$v.bin $n 8 2>&1 | grep -v amdgpu.ids >> $OUT/cdf_probe.log
  done
done
cat $OUT/cdf_probe.log
