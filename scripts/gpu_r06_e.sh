#!/bin/bash
# round 6: the GPU suite (no -x) + the queue-depth variants of rank_match5w_kernel
out=gpurun_out/${1:-r06e}
mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log )
grep -E "passed|failed|FAILED|Error|rc=" $out/pytest_gpu.log | tail -n 12
for v in "" _g8 _h8 _g8h8; do
  timeout 300 scripts/sort5_probe$v.bin 4 4 > $out/sort5_probe$v.log 2>&1; echo "probe$v rc=$?"
  grep -E "rank5w|weighted" $out/sort5_probe$v.log | grep -v adversarial | head -8
  grep -c WRONG $out/sort5_probe$v.log
done
