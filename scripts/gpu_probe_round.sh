# parity tests + sort phase probe (emit and match) + sort micro-benchmarks
TAG=${1:-probe}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOPTEX_SORT_PROBE scripts/sort_phase_probe.hip optimaltextures_amd/csrc/api.hip optimaltextures_amd/csrc/sort_large.hip -o /tmp/sort_probe > $OUT/probe_build.log 2>&1 && timeout 120 /tmp/sort_probe > $OUT/sort_probe.log 2>&1 )
timeout 300 python scripts/microbench.py --only sort --reps 20 > $OUT/sort_micro.log 2>&1
tail -n 3 $OUT/pytest_gpu.log
cat $OUT/probe_build.log $OUT/sort_probe.log $OUT/sort_micro.log
