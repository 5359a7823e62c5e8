# Samples socket power and shader clock while the rotation GEMM / the cdf loop / the sort run back to back
# (is fp32-MFMA throughput clock-limited by the power cap?).  Scratch output under gpurun_out/power/.
mkdir -p gpurun_out/power
export TMPDIR=/tmp
rocm-smi --showpower --showclocks --showmaxpower > gpurun_out/power/idle.txt 2>&1
( for i in $(seq 1 200); do echo "t=$(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk"; sleep 0.2; done ) > gpurun_out/power/samples.txt 2>&1 &
SAMPLER=$!
sleep 2
echo "start_gemm=$(date +%s.%N)" > gpurun_out/power/marks.txt
timeout 120 python scripts/microbench.py --only gemm --reps 3000 > gpurun_out/power/gemm_long.log 2>&1
echo "end_gemm=$(date +%s.%N)" >> gpurun_out/power/marks.txt
sleep 2
echo "start_cdf=$(date +%s.%N)" >> gpurun_out/power/marks.txt
timeout 120 python scripts/microbench.py --only cdf --reps 2000 > gpurun_out/power/cdf_long.log 2>&1
echo "end_cdf=$(date +%s.%N)" >> gpurun_out/power/marks.txt
kill $SAMPLER 2>/dev/null
timeout 300 python scripts/microbench.py --only sort > gpurun_out/power/sort_after_bankfix.log 2>&1
timeout 300 python -m pytest tests -m gpu -x -q -k sort > gpurun_out/power/sort_tests.log 2>&1
cat gpurun_out/power/idle.txt | head -30
cat gpurun_out/power/gemm_long.log gpurun_out/power/sort_after_bankfix.log
tail -3 gpurun_out/power/sort_tests.log
