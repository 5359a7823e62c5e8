mkdir -p gpurun_out/r1
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r1/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r1/pytest_gpu.log )
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r1/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r1/smoke.log )
( timeout 900 python bench.py > gpurun_out/r1/bench_default.log 2>&1; echo "bench rc=$?" >> gpurun_out/r1/bench_default.log )
( timeout 600 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r1/prof -o r1 -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --other_modes "" > gpurun_out/r1/prof_bench.log 2>&1; echo "prof rc=$?" >> gpurun_out/r1/prof_bench.log )
ls -R gpurun_out/r1 | head -40
tail -3 gpurun_out/r1/pytest_gpu.log gpurun_out/r1/smoke.log
tail -2 gpurun_out/r1/bench_default.log
