# throughput vs textures per step (the step is --batch independent textures per GPU)
TAG=${1:-batch}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for b in 16 64; do
  ( timeout 420 python bench.py --batch $b --steps 2 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/bench_b$b.log 2>&1 )
  grep "^{" $OUT/bench_b$b.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('batch', $b, r['value'], 'textures/s', r['ms_per_step'], 'ms/step')"
done
