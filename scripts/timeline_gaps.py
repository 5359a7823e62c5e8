#!/usr/bin/env python3
"""Is a small-batch step GPU-bound, and where are its gaps?  From a `rocprofv3 --kernel-trace -f csv` run of bench.py: per queue
(HIP stream) the busy time of the timed steps, and on the busiest queue the idle time between consecutive kernels, attributed to
the kernel that FOLLOWS the gap (what the queue was waiting to start).
    python scripts/timeline_gaps.py <kernel_trace.csv> [--warmup W] [--out file.md]"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*$", "", name)[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rand = [i for i, r in enumerate(rows) if "distribution_elementwise_grid_stride_kernel" in r["Kernel_Name"]]
    starts = [i for j, i in enumerate(rand) if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[rand[j - 1]]["Start_Timestamp"]) > 2_000_000]
    starts.append(len(rows))
    steps = [(x, y) for x, y in zip(starts[:-1], starts[1:]) if y - x >= 100][a.warmup:]
    # whole steps only (the last one runs to the end of the trace)
    steps = steps[:-1] if len(steps) > 1 else steps
    lines = []
    t_lo, t_hi = int(rows[steps[0][0]]["Start_Timestamp"]), int(rows[steps[-1][1] - 1]["Start_Timestamp"]) if steps[-1][1] < len(rows) else int(rows[-1]["End_Timestamp"])
    sel = [r for x, y in steps for r in rows[x:y]]
    span = (t_hi - t_lo) / 1e6
    lines.append(f"{len(steps)} timed step(s), {len(sel)} kernels, span {span:.2f} ms = {span / len(steps):.2f} ms per step (with the profiler's own gaps)")
    qkey = "Queue_Id" if "Queue_Id" in sel[0] else "Stream_Id"
    byq = collections.defaultdict(list)
    for r in sel:
        byq[r[qkey]].append(r)
    lines.append("")
    lines.append("| queue | kernels | busy ms per step | idle ms per step inside the span |")
    lines.append("|---|---:|---:|---:|")
    main_q, main_busy = None, -1
    for q, rs in byq.items():
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
        lines.append(f"| {q} | {len(rs)} | {busy / len(steps):.2f} | {(span - busy) / len(steps):.2f} |")
        if busy > main_busy:
            main_q, main_busy = q, busy
    rs = byq[main_q]
    gaps = collections.defaultdict(lambda: [0, 0.0])
    hist = collections.Counter()
    for p, r in zip(rs[:-1], rs[1:]):
        g = (int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3   # us
        if g > 0:
            k = short(r["Kernel_Name"])
            gaps[k][0] += 1
            gaps[k][1] += g
            hist["<2us" if g < 2 else "2-5us" if g < 5 else "5-10us" if g < 10 else "10-50us" if g < 50 else ">50us"] += 1
    lines.append("")
    lines.append(f"busiest queue {main_q}: gaps in front of a kernel, by the kernel that follows (top 15 by total), per step")
    lines.append("")
    lines.append("| kernel after the gap | gaps per step | total ms per step | avg us |")
    lines.append("|---|---:|---:|---:|")
    for k, (n, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:15]:
        lines.append(f"| `{k}` | {n / len(steps):.1f} | {tot / 1e3 / len(steps):.3f} | {tot / n:.1f} |")
    lines.append("")
    lines.append("gap sizes: " + ", ".join(f"{k}: {v / len(steps):.0f}" for k, v in hist.items()) + " per step")
    text = "\n".join(lines)
    print(text)
    if a.out:
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
