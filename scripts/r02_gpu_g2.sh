# round 2, GPU call G2: wavefront stagger in the rank step
OUT=gpurun_out/${1:-r02g2}
mkdir -p $OUT
for V in 0 8 16; do for PAD in 0 4096; do echo "stagger $V, LDS pad $PAD"; OPTEX_SORT_LDS_PAD=$PAD timeout 120 scripts/sort_rank4_probe_s$V.bin 16384 12288 2>&1 | grep -E "kernel|rank|queue|total"; done; done | tee $OUT/phases_rank4_stagger.log
export TMPDIR=/tmp
for N in 16384 12544 9216; do
  timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $N --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank4 stag8 n=$N ns=$N /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-230
