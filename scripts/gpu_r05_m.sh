#!/bin/bash
# Round 5, session M: the B = 8 timeline — is the step GPU-bound, where are the gaps
OUT=gpurun_out/r05m
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace -f csv -d $OUT/prof_b8 -o prof -- python bench.py --batch 8 --steps 4 --warmup 2 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/prof_b8.log 2>&1 )
head -1 $OUT/prof_b8/prof_kernel_trace.csv
python scripts/timeline_gaps.py $OUT/prof_b8/prof_kernel_trace.csv --warmup 1 --out $OUT/b8_timeline_gaps.md
python scripts/summarize_rocprof.py $OUT/prof_b8/prof_kernel_trace.csv --warmup 1 --title "bench.py --batch 8, cdf (fused matcher)" --out $OUT/bench_b8_kernel_summary.md > /dev/null 2>&1
head -28 $OUT/bench_b8_kernel_summary.md | cut -c1-150
tail -2 $OUT/prof_b8.log | cut -c1-400
rm -rf $OUT/prof_b8
( timeout 300 python bench.py --batch 8 --steps 8 --warmup 3 --no_cpu_baseline --other_modes "" > $OUT/bench_b8.json 2> $OUT/bench_b8.err )
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05m/bench_b8.json"))
print("B=8 value", d["value"], "ms/step", d["ms_per_step"], "hot", d.get("hot_path_ms_per_step"), "side", d.get("side_stream_ms_per_step"), "other", d.get("other_ms_per_step"))
for k in d["kernels"]:
    print(k["kernel"], k["bound"], k["frac"], k["avg_us"], k["launches"])
PY
