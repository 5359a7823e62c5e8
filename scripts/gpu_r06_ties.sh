#!/bin/bash
# which kernel of the sort chain takes the time on tie-heavy columns: kernel trace of scripts/sort_ties_probe.py, one case per run
out=gpurun_out/${1:-r06ties}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in 1 2; do
  TIES_ONLY=$c timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/trace$c -o t --output-format csv -- python $GRAFT_REPO_ROOT/scripts/sort_ties_probe.py 16384 > $GRAFT_REPO_ROOT/$out/trace$c.log 2>&1
  echo "case $c rc=$?"
  f=$(find $GRAFT_REPO_ROOT/$out/trace$c -name "*kernel_stats.csv" | head -1)
  head -12 $f | cut -c1-200
done
