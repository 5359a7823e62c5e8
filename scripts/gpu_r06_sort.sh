#!/bin/bash
# round 6, sort: rank_match5w_kernel against rank_match4_kernel — timing + checks (sort5_probe), SQ counters of both at 16384 keys
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r06_sort.sh <tag>'
out=gpurun_out/${1:-r06sort}
mkdir -p $out
timeout 400 scripts/sort5_probe.bin 4 8 > $out/sort5_probe.log 2>&1; echo "probe rc=$?"
cd /tmp && export TMPDIR=/tmp
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/$out/pmc_$tag -o pmc --output-format csv -- $GRAFT_REPO_ROOT/scripts/sort5_probe.bin 1 0 16384 > $GRAFT_REPO_ROOT/$out/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_sq.py $(find $out -name "*counter_collection.csv") --match rank_match --elements 268435456 --out $out/sort_sq_counters.md --title "rank_match5w_kernel vs rank_match4_kernel: SQ counters, [64 x 256] columns of 16384 keys" 2>&1 | tail -3
cat $out/sort_sq_counters.md | grep -v "^$" | head -60
grep -v adversarial $out/sort5_probe.log; grep -c WRONG $out/sort5_probe.log
