# sort_rank2: parity tests for the sort mode + phase probe + micro-benchmarks, new kernel vs the one-column-per-CU kernel
TAG=${1:-sort2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 300 python -m pytest tests -m gpu -x -q -k "sort or ot_loop or smoke or driver" > $OUT/tests_sort.log 2>&1; echo "pytest rc=$?" >> $OUT/tests_sort.log )
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOPTEX_SORT_PROBE scripts/sort_rank2_probe.hip optimaltextures_amd/csrc/api.hip -o /tmp/r2probe 2>/dev/null && timeout 120 /tmp/r2probe > $OUT/probe.log 2>&1 )
for n in 16384 12544 9216 6400 4096; do
  timeout 300 python scripts/microbench.py --only sort --reps 20 --n $n --ns 12288 >> $OUT/sort_rank2.log 2>&1
  if [ -n "$WITH_RANK1" ]; then OPTEX_SORT_PATH=rank1 timeout 300 python scripts/microbench.py --only sort --reps 20 --n $n --ns 12288 >> $OUT/sort_rank1.log 2>&1; fi
done
tail -n 3 $OUT/tests_sort.log
cat $OUT/probe.log
echo "--- rank2"; grep '"kernel": "sort_match"' $OUT/sort_rank2.log
