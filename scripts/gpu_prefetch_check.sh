export TMPDIR=/tmp
mkdir -p gpurun_out/prefetch
timeout 300 python -m pytest tests -m gpu -x -q -k "driver or config or prefetch or codec" 2>&1 | tail -4
( timeout 400 python bench.py --steps 2 --no_cpu_baseline --other_modes "" --no_kernel_timing > gpurun_out/prefetch/bench.log 2>&1 ); grep "^{" gpurun_out/prefetch/bench.log | cut -c1-200
