OUT=gpurun_out/${1:-r02n2}
mkdir -p $OUT
export TMPDIR=/tmp
for T in 2 1; do echo "OPTEX_GLUE_TP128=$T"; ( OPTEX_GLUE_TP128=$T timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "glue or vgg or codec" 2>&1 | tail -1 ); OPTEX_GLUE_TP128=$T timeout 300 python scripts/glue_layout_bench.py 32 2>/dev/null; done 2>&1 | tee $OUT/glue_layout_bench.log
