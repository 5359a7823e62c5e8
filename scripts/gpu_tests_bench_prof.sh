# Full GPU validation: parity tests, smoke, default bench (JSON line), rocprofv3 kernel trace of a short bench run.
# Usage (from the repo root, on the GPU box):  bash scripts/gpu_tests_bench_prof.sh <tag>
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 420 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log )
( timeout 600 python bench.py > $OUT/bench_default.log 2>&1; echo "bench rc=$?" >> $OUT/bench_default.log )

( timeout 480 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --other_modes "" > $OUT/prof_bench.log 2>&1; echo "prof rc=$?" >> $OUT/prof_bench.log )
tail -n 3 $OUT/pytest_gpu.log $OUT/smoke.log
tail -n 2 $OUT/bench_default.log

