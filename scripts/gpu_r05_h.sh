#!/bin/bash
# Round 5, session H: the fused cdf matcher (column in registers) against the two-kernel pipeline — parity tests, the loop
# micro-benchmark at the pass sizes with both settings, a short bench
OUT=gpurun_out/r05h
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x -k "cdf or ot_loop or hist or optimal_transport or forward or abi" 2>&1 | tail -15 ) > $OUT/pytest_cdf.log 2>&1
tail -5 $OUT/pytest_cdf.log
for n in 16384 9216 4096; do
  for f in 1 0; do
    ( timeout 300 python scripts/microbench.py --S 64 --n $n --ns $((n * 3 / 4)) --only loop,cdf --cdf_fused $f ) >> $OUT/microbench_cdf.log 2>&1
  done
done
grep -v "gemm_tn\|col_minmax" $OUT/microbench_cdf.log | cut -c1-200
( timeout 900 python bench.py --steps 3 --warmup 2 --other_modes batch8 --no_cpu_baseline > $OUT/bench_short.json 2> $OUT/bench_short.err; echo "rc=$?" >> $OUT/bench_short.err )
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05h/bench_short.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "batch8", d.get("textures_per_s_batch8"))
    print("hot", d.get("hot_path_ms_per_step"), "side", d.get("side_stream_ms_per_step"), "other", d.get("other_ms_per_step"))
    for k in d["kernels"]:
        print(k["kernel"], k["bound"], k["frac"], k["avg_us"], k["launches"])
except Exception as e:
    print("bench failed", e)
PY
tail -3 $OUT/bench_short.err
