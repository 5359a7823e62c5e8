# layout-aware glue + mixed-layout codec: parity tests, then the default bench in both layout policies
TAG=${1:-glue2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
( OPTEX_CODEC_LAYOUT=nchw timeout 600 python bench.py --no_cpu_baseline --other_modes "" > $OUT/bench_nchw.log 2>&1 )
( OPTEX_CODEC_LAYOUT=mixed timeout 600 python bench.py --no_cpu_baseline --other_modes "" > $OUT/bench_mixed.log 2>&1 )
tail -n 4 $OUT/pytest_gpu.log
for f in nchw mixed; do python - <<PY
import json
l=[x for x in open("$OUT/bench_$f.log") if x.startswith("{")]
if l:
    r=json.loads(l[-1]); print("$f", r["value"], "textures/s", r["ms_per_step"], "ms/step hot", r.get("hot_path_ms_per_step"), "other", r.get("other_ms_per_step"), [ (k["kernel"], k["avg_us"], k["frac"]) for k in r["kernels"] if k["kernel"]=="vgg_glue"])
else:
    print("$f: no result"); print(open("$OUT/bench_$f.log").read()[-2000:])
PY
done
