#!/bin/bash
# Round 5, session C: LDS-staged matrix prologue of the R-stationary GEMM, persistent sort kernel, histogram binning without the
# division sequence, 4-row glue kernel: parity (whole GPU suite), probes, a short bench.
OUT=gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $OUT/pytest_gpu.log 2>&1
tail -8 $OUT/pytest_gpu.log
( scripts/sort_time_probe.bin 4 ) > $OUT/sort_time_probe.log 2>&1
cat $OUT/sort_time_probe.log
P=scripts/gemm_timeline_probe.bin
( $P $OUT/tl_rs_b64.bin 0 64 16384 0 0
  $P $OUT/tl_rs_b64_rowstat.bin 0 64 16384 1 0
  for N in 4096 9216; do $P $OUT/tl_rs_b8_$N.bin 0 8 $N 1 0; done ) > $OUT/timeline_probe.log 2>&1
cat $OUT/timeline_probe.log
python scripts/gemm_timeline_report.py $OUT/tl_rs_b64.bin $OUT/tl_rs_b64_rowstat.bin $OUT/tl_rs_b8_4096.bin $OUT/tl_rs_b8_9216.bin > $OUT/gemm_timeline.md 2> $OUT/report.err
rm -f $OUT/tl_*.bin
( for N in 4096 6400 9216 12544 16384; do scripts/gemm_rs_probe_d16.bin 8 $N 256 256 50 0 0; scripts/gemm_rs_probe_d16.bin 8 $N 256 256 50 1 0; done
  scripts/gemm_rs_probe_d16.bin 64 16384 256 256 20 0 0; scripts/gemm_rs_probe_d16.bin 64 16384 256 256 20 1 0
  scripts/gemm_rs_probe_d16.bin 64 4096 256 256 20 0 0; scripts/gemm_rs_probe_d16.bin 64 4096 256 256 20 1 0
  scripts/gemm_rs_probe_d16.bin 64 16384 181 181 20 0 0; scripts/gemm_rs_probe_d16.bin 64 16384 128 128 20 0 0 ) > $OUT/gemm_probe.log 2>&1
cat $OUT/gemm_probe.log
( timeout 600 python scripts/batch_probe.py 8 64 ) > $OUT/batch_probe.log 2>&1
tail -3 $OUT/batch_probe.log | cut -c1-600
( timeout 900 python bench.py --steps 3 --warmup 2 --other_modes sort,batch8 --no_cpu_baseline > $OUT/bench_short.json 2> $OUT/bench_short.err; echo "rc=$?" >> $OUT/bench_short.err )
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05c/bench_short.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "batch8", d.get("textures_per_s_batch8"), "by mode", d.get("textures_per_s_by_hist_mode"))
    for k in d["kernels"]:
        print(k["kernel"], k["bound"], k["frac"], k["avg_us"], k["launches"])
    print("sort", [(k["kernel"], k["frac"], k["avg_us"]) for k in d.get("sort_kernels", [])])
except Exception as e:
    print("bench failed", e)
PY
grep -v "^| k-steps [0-9]*\.\.[0-9]* " $OUT/gemm_timeline.md | grep "^## \|entry ->\|issued ->\|whole tile\|epilogue\|effective" | cut -c1-200
