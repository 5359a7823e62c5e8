OUT=gpurun_out/${1:-r02d2}
mkdir -p $OUT
export TMPDIR=/tmp
for N in 16384 12544 9216 6400 4096; do
  timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $N --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 ${2:-variant} n=$N ns=$N /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-230
