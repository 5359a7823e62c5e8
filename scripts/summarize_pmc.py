#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes over bench.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; separate runs as
the MI355X guide prescribes) into profiles/<name>.json: average HBM-side bytes per launch for every optex kernel class
in the TIMED steps.

    python scripts/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> --out profiles/x.json

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB... the counters report kilobytes
(x 1024 here); on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes for wide coalesced streaming reads, so it is
DOUBLED; WRITE_SIZE matched known byte counts in our own access patterns (checked on the rotation GEMM, the LUT apply
and the sort: written bytes == output size) and is used as is.  Infinity-cache hits are included in both.
"""
import argparse
import collections
import csv
import json
import re

CLASS_OF = [("gemm_tn_kernel", "gemm_tn"), ("gemm16_cm_kernel", "gemm_tn"), ("gemm_rs_kernel", "gemm_tn"), ("cdf_apply_kernel", "cdf_apply"), ("cdf_fused_kernel", "cdf_match"),
            ("col_hist_kernel", "col_hist"), ("cdf_hist_lut_kernel", "col_hist"), ("col_minmax_kernel", "col_minmax"), ("cdf_lut_kernel", "cdf_lut"),
            ("mt_accept_kernel", "legacy_normals"), ("normals_emit_kernel", "legacy_normals"), ("gram_tri_kernel", "gram"),
            ("chol_inv2_kernel", "chol_inv"),
            ("glue_kernel", "vgg_glue"), ("glue_nhwc_kernel", "vgg_glue"), ("glue_transpose", "vgg_glue"),
            ("rank_match4_kernel", "sort_rank4"),
            ("gram128_kernel", "gram"), ("minmax_from_parts_kernel", "col_minmax"), ("mean_from_parts_kernel", "col_mean"),
            ("chol_inv_kernel", "chol_inv"), ("ns_init_kernel", "ns_init"), ("cov_finalize_kernel", "cov_finalize"), ("sort_columns_kernel", "sort_radix"), ("gram_kernel", "gram"),
            ("col_mean_kernel", "col_mean"), ("householder_apply", "householder")]


def classify(name):
    for key, cls in CLASS_OF:
        if key in name:
            if cls == "sort_rank4":  # rank_match4_kernel<ITEMS, VEC, NT, FULL, MODE>: MODE 1 = match, 0 = emit (sort_columns)
                m = re.search(r"rank_match4_kernel<[^>]*, (\d)>", name)
                return "sort_columns" if m and m.group(1) == "0" else "sort_match"
            return cls
    return None


def timed_rows(path, warmup):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps = step_intervals(rows)[warmup:]
    return [r for a, b in steps for r in rows[a:b]], len(steps)


def step_intervals(rows):
    """A bench step starts with the torch.rand() launches that create its pastiche batch (`distribution_elementwise...`, one per
    texture, microseconds apart) and holds hundreds of kernels.  Since round 4 the rotation generator works on its own stream,
    AHEAD of the main stream, so its kernels (householder_prep, ...) may carry timestamps of the previous step: a step is the
    span from the first rand launch of a burst (> 2 ms after the previous rand launch) to the first of the next burst."""
    rand = [i for i, r in enumerate(rows) if "distribution_elementwise_grid_stride_kernel" in r["Kernel_Name"]]
    starts = [i for j, i in enumerate(rand)
              if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[rand[j - 1]]["Start_Timestamp"]) > 2_000_000]
    starts.append(len(rows))
    return [(a, b) for a, b in zip(starts[:-1], starts[1:]) if b - a >= 100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default="")
    ap.add_argument("--measured", default="", help="what / when this was measured on (round, commit), copied into the file")
    args = ap.parse_args()
    out = {"command": args.command, "measured": args.measured, "units": "bytes per launch, averaged over the launches of the timed steps",
           "corrections": {"FETCH_SIZE": "x1024 x2 (gfx950 counts 128-B requests as 64 B on wide streaming reads)",
                           "WRITE_SIZE": "x1024"}, "kernels": {}}
    for path, counter, scale in ((args.fetch_csv, "FETCH_SIZE", 2048.0), (args.write_csv, "WRITE_SIZE", 1024.0)):
        rows, nsteps = timed_rows(path, args.warmup)
        agg = collections.defaultdict(list)
        for r in rows:
            if r["Counter_Name"] != counter:
                continue
            cls = classify(r["Kernel_Name"])
            if cls:
                agg[cls].append(float(r["Counter_Value"]) * scale)
        for cls, vals in agg.items():
            e = out["kernels"].setdefault(cls, {})
            e[counter.lower() + "_bytes"] = sum(vals) / len(vals)
            e["launches_" + counter.lower()] = len(vals)
        out["timed_steps"] = nsteps
    for cls, e in out["kernels"].items():
        e["hbm_bytes"] = e.get("fetch_size_bytes", 0.0) + e.get("write_size_bytes", 0.0)
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
