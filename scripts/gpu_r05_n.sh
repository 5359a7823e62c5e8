#!/bin/bash
# Round 5, session N: fed generator schedule + auto spare CUs at B = 8; gemm_rs with centring at M = 256 (chol row)
OUT=gpurun_out/r05n
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x -k "fed_during or step_ahead or rs_gemm or linear or chol or device_stream or hist_match_linear or determinism" 2>&1 | tail -8 ) > $OUT/pytest_sel.log 2>&1
tail -4 $OUT/pytest_sel.log
( timeout 600 python scripts/batch_probe.py 8 ) > $OUT/batch_probe.log 2>&1
tail -2 $OUT/batch_probe.log | tr '|' '\n' | cut -c1-200
( timeout 900 python bench.py --steps 3 --warmup 2 --other_modes chol,batch8 --no_cpu_baseline > $OUT/bench_short.json 2> $OUT/bench_short.err; echo "rc=$?" >> $OUT/bench_short.err )
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05n/bench_short.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "batch8", d.get("textures_per_s_batch8"), "by mode", d.get("textures_per_s_by_hist_mode"), "literal", d.get("textures_per_s_literal_linear_sequence"))
    print("hot", d.get("hot_path_ms_per_step"), "side", d.get("side_stream_ms_per_step"), "other", d.get("other_ms_per_step"))
    for k in d["kernels"]:
        print(k["kernel"], k["bound"], k["frac"], k["avg_us"], k["launches"])
except Exception as e:
    print("bench failed", e)
PY
tail -3 $OUT/bench_short.err
