# rank_match_kernel: effect of R2_TMAX (longest run ranked in the G-wide loop) on the [32 x 256 x 16384] batch
export TMPDIR=/tmp
mkdir -p gpurun_out/tmax
for t in 4 5 6 7 8; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOPTEX_SORT_PROBE -DR2_TMAX_VALUE=$t scripts/sort_rank2_probe.hip optimaltextures_amd/csrc/api.hip -o /tmp/r2probe_$t 2>/dev/null && echo "TMAX=$t: $(timeout 60 /tmp/r2probe_$t | grep '8192 columns')"
done | tee gpurun_out/tmax/tmax.log
timeout 200 python -m pytest tests -m gpu -x -q -k "sort" 2>&1 | tail -2
