#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats -f csv` run of bench.py (or scripts/microbench.py) into the per-kernel
summary committed under profiles/.

    python scripts/summarize_rocprof.py gpurun_out/<dir>/<prefix>_kernel_trace.csv [--warmup W] [--out profiles/x.md]

bench.py's warm-up step runs MIOpen's find mode (hundreds of trial convolutions), which would swamp a whole-run
--stats table; this script therefore keeps only the TIMED steps: a step starts at the torch.rand() launch that creates
its pastiche batch (`distribution_elementwise_grid_stride_kernel`, a burst of one launch per texture) and holds at least 100
kernels.  With --all the whole trace is summarised instead (micro-benchmarks).
"""
import argparse
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*$", "", name)   # drop the trailing argument list
    name = name.replace("at::native::", "")
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--title", type=str, default="")
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n_steps = 0
    if not args.all:
        # a step = from the first torch.rand() launch of a burst (one launch per texture, microseconds apart; > 2 ms after the
        # previous rand launch) to the first of the next burst, at least 100 kernels.  (Up to round 3 a step was recognised by
        # its >= 5 householder_prep launches; the rotation generator now runs AHEAD on its own stream, so those may carry
        # timestamps of the previous step.)
        rand = [i for i, r in enumerate(rows) if "distribution_elementwise_grid_stride_kernel" in r["Kernel_Name"]]
        starts = [i for j, i in enumerate(rand)
                  if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[rand[j - 1]]["Start_Timestamp"]) > 2_000_000]
        starts.append(len(rows))
        steps = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if b - a >= 100]
        steps = steps[args.warmup:]
        if not steps:
            sys.exit("no timed steps found in the trace")
        # the last step runs to the end of the trace: cut it after its last optex/VGG kernel (the isfinite check follows)
        rows = [r for a, b in steps for r in rows[a:b]]
        n_steps = len(steps)
    agg = collections.OrderedDict()
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = short(r["Kernel_Name"])
        e = agg.setdefault(k, dict(n=0, tot=0, mn=1 << 62, mx=0, vgpr=r["VGPR_Count"], agpr=r["Accum_VGPR_Count"],
                                   lds=r["LDS_Block_Size"], wg=r["Workgroup_Size_X"]))
        e["n"] += 1
        e["tot"] += d
        e["mn"] = min(e["mn"], d)
        e["mx"] = max(e["mx"], d)
    busy = sum(e["tot"] for e in agg.values())
    span = max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)
    lines = []
    if args.title:
        lines.append(f"# {args.title}\n")
    lines.append(f"source: `{args.trace}`  ({'whole trace' if args.all else f'{n_steps} timed step(s), warm-up skipped'})\n")
    lines.append(f"kernel busy time {busy / 1e6:.2f} ms over a span of {span / 1e6:.2f} ms (the span includes profiler gaps)\n")
    lines.append("| kernel | calls | total ms | avg us | min us | max us | % busy | VGPR+AGPR | LDS B | WG |")
    lines.append("|---|---:|---:|---:|---:|---:|---:|---|---:|---:|")
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        if e["tot"] / busy < 0.0005 and not k.startswith("optex"):
            continue
        lines.append(f"| `{k}` | {e['n']} | {e['tot'] / 1e6:.3f} | {e['tot'] / e['n'] / 1e3:.1f} | {e['mn'] / 1e3:.1f} | "
                     f"{e['mx'] / 1e3:.1f} | {100 * e['tot'] / busy:.1f} | {e['vgpr']}+{e['agpr']} | {e['lds']} | {e['wg']} |")
    ours = sum(e["tot"] for k, e in agg.items() if k.startswith("optex"))
    lines.append(f"\noptex:: kernels: {ours / 1e6:.2f} ms = {100 * ours / busy:.1f} % of kernel busy time")
    text = "\n".join(lines) + "\n"
    if args.out:
        open(args.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
