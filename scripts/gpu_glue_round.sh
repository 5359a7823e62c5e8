mkdir -p gpurun_out/glue
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "glue or codec or fit_pca or mix_style or driver_loop" > gpurun_out/glue/tests.log 2>&1
timeout 300 python scripts/microbench.py --only glue > gpurun_out/glue/microbench.log 2>&1
timeout 600 python scripts/conv_layout_probe.py > gpurun_out/glue/conv_layout.log 2>&1
timeout 900 python bench.py --other_modes "" > gpurun_out/glue/bench.log 2>&1
tail -n 5 gpurun_out/glue/tests.log
cat gpurun_out/glue/microbench.log gpurun_out/glue/conv_layout.log
tail -n 1 gpurun_out/glue/bench.log
