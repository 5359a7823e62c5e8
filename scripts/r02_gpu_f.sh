# round 2, GPU call F: owner-ranked sort match kernel after the ILP restructure: parity, microbench, per-phase probe
OUT=gpurun_out/${1:-r02f}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_sort.log | tail -n 20
for N in 16384 12544 9216 6400 4096; do
  NS=$((N*3/4))
  OPTEX_SORT_PATH=rank3 timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $NS --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank3 n=$N /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-200
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOPTEX_SORT_PROBE scripts/sort_rank3_probe.hip optimaltextures_amd/csrc/api.hip -o /tmp/sort3_probe > $OUT/probe_build.log 2>&1
( timeout 120 /tmp/sort3_probe 16384 12288; timeout 120 /tmp/sort3_probe 9216 6912 ) | tee $OUT/probe.log
