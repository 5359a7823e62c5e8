# round 2, GPU call A2: keys per thread vs workgroup size for columns up to 8192 keys
OUT=gpurun_out/${1:-r02a2}
mkdir -p $OUT
export TMPDIR=/tmp
for E in 3 2 1; do
for N in 4096 6400 8192 3072 5120; do
  OPTEX_SORT_EXTRA_NT=$E timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $N --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 extra_nt=$E n=$N /"
done; done | tee $OUT/microbench_sortmatch.log | cut -c1-230
