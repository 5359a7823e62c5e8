#!/usr/bin/env python3
"""Copy what scripts/gpu_evidence.sh left under gpurun_out/<tag>/ into profiles/ under round-stamped names and merge the two
PMC traffic passes (cdf kernels from the cdf run, sort kernels from the sort run) into profiles/pmc_traffic.json, the file
bench.py reads for `roofline.traffic`.      python scripts/collect_profiles.py <tag> <prefix, e.g. r03>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pre = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for name in ("bench_default.json", "bench_b64_cdf_kernel_summary.md", "bench_b64_sort_kernel_summary.md",
             "bench_b64_chol_kernel_summary.md", "bench_b64_sym_kernel_summary.md", "bench_b64_pca_kernel_summary.md",
             "bench_b8_cdf_kernel_summary.md", "bench_b64_ownrotations_kernel_summary.md",
             "single_texture_kernel_summary.md", "gemm_mfma_counters.md", "sort_match4_sq_counters.md",
             "sort_rank4_phases.log", "sort_time_probe.log", "sort_columns_microbench.log", "normals_probe.log",
             "batch_probe.log", "gram_probe.md", "ns_count_probe.md", "cdf_probe.log", "colcopy_probe.log", "b8_timeline_gaps.md",
             "b8_layout_probe.log", "cdf_fused_sq_counters.md", "sort_match5w_sq_counters.md", "sort5_probe.log", "chol_probe.log",
             "sort_stress_loop.log", "sort_ties_probe.log", "glue_planar_probe.log", "pytest_gpu.log"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{pre}_{name}"))
if not os.path.exists(os.path.join(src, "pmc_traffic_cdf.json")):
    sys.exit(0)   # a session without PMC passes: the files above are all there is to collect
cdf = json.load(open(os.path.join(src, "pmc_traffic_cdf.json")))
srt = json.load(open(os.path.join(src, "pmc_traffic_sort.json")))
for n, d in (("cdf", cdf), ("sort", srt)):
    json.dump(d, open(os.path.join(dst, f"{pre}_pmc_traffic_bench_b64_{n}.json"), "w"), indent=1, sort_keys=True)
merged = dict(cdf)
merged["kernels"] = dict(cdf["kernels"])
for k, v in srt["kernels"].items():
    if k.startswith("sort"):
        merged["kernels"][k] = v
merged["measured"] = (f"{cdf.get('measured', '')}: cdf-mode kernels from profiles/{pre}_pmc_traffic_bench_b64_cdf.json, sort kernels from "
                      f"profiles/{pre}_pmc_traffic_bench_b64_sort.json (the same command with --hist_mode sort)")
json.dump(merged, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print({k: round(v["hbm_bytes"] / 1e6, 1) for k, v in merged["kernels"].items()})
