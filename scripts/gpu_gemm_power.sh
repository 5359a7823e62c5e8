# Rotation GEMM at the hot-loop shape, both hot kernels, three kinds of data, un-profiled and under rocprofv3 --pmc: the source of
# profiles/r03_gemm_power_dvfs.md (probe binaries: make -C scripts).   gpurun -- bash scripts/gpu_gemm_power.sh
OUT=gpurun_out/r03_gemm_power
mkdir -p $OUT
export TMPDIR=/tmp
echo "== un-profiled (HIP events) ==" > $OUT/summary.log
for old in 0 1; do for mode in 0 2 1; do scripts/gemm_rs_probe_d16.bin 64 16384 256 256 20 $old $mode >> $OUT/summary.log; done; done
scripts/gemm_rs_probe_noload.bin 64 16384 256 256 20 0 0 >> $OUT/summary.log
for mk in "181 181" "192 192" "181 256" "256 181"; do for old in 0 1; do scripts/gemm_rs_probe_d16.bin 64 16384 $mk 10 $old 0 >> $OUT/summary.log; done; done
for n in 4096 6400 9216 12544; do for old in 0 1; do scripts/gemm_rs_probe_d16.bin 64 $n 256 256 20 $old 0 >> $OUT/summary.log; done; done
echo "== rocprofv3 --pmc (clock = GRBM_GUI_ACTIVE / 8 / duration; MFMA utilisation) ==" >> $OUT/summary.log
for old in 0 1; do for mode in 0 2 1; do
  v=old${old}_mode${mode}
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -f csv -d $OUT/$v -o pmc -- scripts/gemm_rs_probe_d16.bin 64 16384 256 256 6 $old $mode > $OUT/$v.log 2>&1
  python scripts/summarize_sq.py $OUT/$v/pmc_counter_collection.csv --match gemm --skip 3 --out $OUT/$v.md --title "$v" > /dev/null 2>&1
  echo "-- $v: $(grep -h 'data:' $OUT/$v.log)" >> $OUT/summary.log
  grep -E "^## |^\* MFMA|^\* eff|duration" $OUT/$v.md >> $OUT/summary.log
  rm -rf $OUT/$v
done; done
cat $OUT/summary.log
