#!/bin/bash
# Round 5, session G: whole GPU suite on the routed library; short bench (cdf + sort + chol + batch8); B = 8 probe; convolution instances at B = 8 and B = 64
OUT=gpurun_out/r05g
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $OUT/pytest_gpu.log 2>&1
tail -8 $OUT/pytest_gpu.log
( timeout 900 python bench.py --steps 3 --warmup 2 --other_modes sort,chol,batch8 --no_cpu_baseline > $OUT/bench_short.json 2> $OUT/bench_short.err; echo "rc=$?" >> $OUT/bench_short.err )
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05g/bench_short.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "batch8", d.get("textures_per_s_batch8"), "by mode", d.get("textures_per_s_by_hist_mode"), "literal", d.get("textures_per_s_literal_linear_sequence"))
    print("hot", d.get("hot_path_ms_per_step"), "side", d.get("side_stream_ms_per_step"), "other", d.get("other_ms_per_step"))
    for k in d["kernels"]:
        print(k["kernel"], k["bound"], k["frac"], k["avg_us"], k["launches"])
    print("sort", [(k["kernel"], k["frac"], k["avg_us"]) for k in d.get("sort_kernels", [])])
except Exception as e:
    print("bench failed", e)
PY
tail -3 $OUT/bench_short.err
( timeout 600 python scripts/batch_probe.py 8 64 ) > $OUT/batch_probe.log 2>&1
tail -3 $OUT/batch_probe.log | cut -c1-600
for B in 8 64; do
  ( timeout 600 rocprofv3 --kernel-trace -f csv -d $OUT/prof_b$B -o prof -- python bench.py --batch $B --steps 2 --warmup 1 --no_cpu_baseline --other_modes "" --no_kernel_timing > $OUT/prof_b$B.log 2>&1 )
  python scripts/summarize_rocprof.py $OUT/prof_b$B/prof_kernel_trace.csv --warmup 1 --title "bench.py --batch $B, cdf (round 5 work in progress)" --out $OUT/bench_b${B}_kernel_summary.md > /dev/null 2>&1
  python scripts/conv_instances.py $OUT/prof_b$B/prof_kernel_trace.csv 30 > $OUT/conv_instances_b$B.md 2>&1
  rm -rf $OUT/prof_b$B
done
head -30 $OUT/bench_b8_kernel_summary.md | cut -c1-170
head -24 $OUT/conv_instances_b8.md | cut -c1-170
