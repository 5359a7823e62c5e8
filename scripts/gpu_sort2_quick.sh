# quick: sort parity tests + match micro-benchmark at the five pass sizes
export TMPDIR=/tmp
mkdir -p gpurun_out/sortq
timeout 200 python -m pytest tests -m gpu -x -q -k "sort" 2>&1 | tail -2
for n in 16384 12544 9216 6400 4096; do
  timeout 120 python scripts/microbench.py --only sort --reps 20 --n $n --ns 12288 2>/dev/null | grep '"kernel": "sort_match"' | grep '_sort_match"'
done | tee gpurun_out/sortq/micro.log
