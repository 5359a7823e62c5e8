# counters of the R-stationary GEMM probe variants
OUT=gpurun_out/r03_rspmc
mkdir -p $OUT
export TMPDIR=/tmp
for v in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -f csv -d $OUT/$v -o pmc -- scripts/gemm_rs_probe_$v.bin 64 16384 256 256 6 > $OUT/$v.log 2>&1
  python scripts/summarize_sq.py $OUT/$v/pmc_counter_collection.csv --match gemm_rs --skip 3 --out $OUT/$v.md --title "gemm_rs $v" > /dev/null 2>&1
  grep -E "^\* |duration" $OUT/$v.md
  rm -rf $OUT/$v
done
