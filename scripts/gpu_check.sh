#!/bin/bash
# One GPU-box session: the -m gpu suite, smoke(), and the default bench line.  Everything lands under gpurun_out/<tag>/.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_check.sh <tag> [pytest-args]'
TAG=${1:-check}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 2000 python -m pytest tests -m gpu -q -x "$@" > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log )
grep -E "passed|failed|FAILED|Error|rc=" $OUT/pytest_gpu.log | tail -n 12
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log ); tail -2 $OUT/smoke.log
( timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -3 $OUT/bench_default.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    keep = {k: d.get(k) for k in ("value", "ms_per_step", "roofline", "textures_per_s_by_hist_mode", "textures_per_s_pca_default",
                                  "textures_per_s_independent_rotations", "hot_path_ms_per_step", "other_ms_per_step")}
    print(json.dumps(keep, indent=1))
    for k in d.get("kernels", []):
        print(k["kernel"], k["achieved"], k["unit"], k["frac"], k["avg_us"], k["launches"])
    for k in d.get("sort_kernels", []):
        print("sort:", k["kernel"], k["achieved"], k["unit"], k["frac"], k["avg_us"], k["launches"])
    print(d.get("cpu_baseline"))
except Exception as e:
    print("bench parse failed:", e)
PY
