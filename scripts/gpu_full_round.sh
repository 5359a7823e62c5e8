TAG=${1:-r1c}
bash scripts/gpu_tests_bench_prof.sh $TAG
bash scripts/gpu_pmc_traffic.sh ${TAG}_pmc
