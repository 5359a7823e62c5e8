# last check of a round: full GPU test suite, smoke, default bench (no profiler)
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 420 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log )
( time timeout 600 python bench.py > $OUT/bench_default.log 2>&1; echo "bench rc=$?" >> $OUT/bench_default.log ) 2> $OUT/bench_time.log
tail -n 3 $OUT/pytest_gpu.log $OUT/smoke.log
tail -n 2 $OUT/bench_default.log | cut -c1-400
cat $OUT/bench_time.log
