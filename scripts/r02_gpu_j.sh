# round 2, GPU call J: sort rank3 after scan/queue trims (parity, microbench, probe, SQ counters); linear modes with the
# apply and the rotation back as one GEMM (linalg tests, forward fixtures incl. the teacher-forced five-layer case), bench
OUT=gpurun_out/${1:-r02j}
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sort" > $OUT/pytest_sort.log 2>&1; echo "rc=$?" >> $OUT/pytest_sort.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_sort.log | tail -n 5
for N in 16384 12544 9216 6400 4096; do
  NS=$((N*3/4))
  timeout 300 python scripts/microbench.py --only sortmatch --S 64 --n $N --ns $NS --reps 10 2>/dev/null | grep '"kernel": "sort_match"' | sed "s/^/rank3 n=$N /"
done | tee $OUT/microbench_sortmatch.log | cut -c1-200
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOPTEX_SORT_PROBE scripts/sort_rank3_probe.hip optimaltextures_amd/csrc/api.hip -o /tmp/sort3_probe > $OUT/probe_build.log 2>&1
( timeout 120 /tmp/sort3_probe 16384 12288 ) | tee $OUT/probe.log
( timeout 900 python -m pytest tests/test_gpu_linalg.py -m gpu -q > $OUT/pytest_linalg.log 2>&1; echo "rc=$?" >> $OUT/pytest_linalg.log )
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_linalg.log | tail -n 8
( timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "forward_matches" > $OUT/pytest_forward.log 2>&1; echo "rc=$?" >> $OUT/pytest_forward.log )
grep -E "passed|failed|FAILED|max err|rc=" $OUT/pytest_forward.log | tail -n 20
( timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err )
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d.get('textures_per_s_by_hist_mode'), d.get('textures_per_s_fused_by_hist_mode'), d.get('textures_per_s_reference_defaults')); print([ (k['kernel'],k['frac'],k['avg_us']) for k in d.get('sort_kernels',[])])"
tail -3 $OUT/bench.err
MB="python scripts/microbench.py --only sortmatch --S 64 --reps 6"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $OUT/sort_sq1 -o pmc -- $MB > $OUT/sort_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -f csv -d $OUT/sort_sq2 -o pmc -- $MB > $OUT/sort_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN -f csv -d $OUT/sort_sq3 -o pmc -- $MB > $OUT/sort_sq3.log 2>&1
python scripts/summarize_sq.py $OUT/sort_sq1/pmc_counter_collection.csv $OUT/sort_sq2/pmc_counter_collection.csv $OUT/sort_sq3/pmc_counter_collection.csv --match rank_match3 --skip 3 --elements $((64*256*16384)) --title "rank_match3_kernel ([64, 256, 16384] against a [1, 256, 12288] style): instruction mix and wait states" --command "rocprofv3 --kernel-trace --pmc <counters> -- $MB" --out $OUT/sort_match3_sq_counters.md > /dev/null 2>&1
rm -rf $OUT/sort_sq1 $OUT/sort_sq2 $OUT/sort_sq3
tail -n 14 $OUT/sort_match3_sq_counters.md
