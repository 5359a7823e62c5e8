# round 2, GPU call M2: 128-pixel tiles in the layout-changing glue kernel: parity (glue tests), layout bench both ways
OUT=gpurun_out/${1:-r02m2}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "glue or vgg or codec" > $OUT/pytest_glue.log 2>&1; echo "rc=$?" >> $OUT/pytest_glue.log )
grep -E "passed|failed|FAILED|rc=|Error|error" $OUT/pytest_glue.log | tail -n 6
for T in 1 0; do echo "OPTEX_GLUE_TP128=$T"; OPTEX_GLUE_TP128=$T timeout 300 python scripts/glue_layout_bench.py 32; done 2>&1 | tee $OUT/glue_layout_bench.log
