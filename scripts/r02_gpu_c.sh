# round 2, GPU call C: linear-mode kernels after the Newton-Schulz symmetry fix (tests without -x), forward fixtures, CLI,
# microbench of the linalg kernels, then the measurement passes (scripts/r02_gpu_measure.sh)
OUT=gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_linalg.py -m gpu -q -s --durations=5 > $OUT/pytest_linalg.log 2>&1; echo "rc=$?" >> $OUT/pytest_linalg.log )
( timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "forward_matches or cli" > $OUT/pytest_forward.log 2>&1; echo "rc=$?" >> $OUT/pytest_forward.log )
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q > $OUT/pytest_parity.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity.log )
( timeout 600 python scripts/microbench.py --only linalg --S 64 > $OUT/microbench_linalg.log 2>&1; echo "rc=$?" >> $OUT/microbench_linalg.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_linalg.log | tail -n 20
grep -E "passed|failed|FAILED|max err|rc=|PCA ranks" $OUT/pytest_forward.log | tail -n 20
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_parity.log | tail -n 8
cat $OUT/microbench_linalg.log | cut -c1-220
bash scripts/r02_gpu_measure.sh r02c_measure
