#!/usr/bin/env python3
"""For every launch of the kernels matching PATTERN in a rocprofv3 kernel trace: its duration, grid, and which kernels of OTHER
queues ran at the same time (share of the launch's duration they overlap).  Answers "is this kernel slower inside the bench
than alone because something runs beside it?".
    python scripts/overlap_report.py <kernel_trace.csv> PATTERN [max rows]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"],
                         (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))))
    rows.sort()
    print("| kernel | grid (threads) | us | beside it (other queues: share of this launch's duration) |")
    print("|---|---|---:|---|")
    shown = 0
    for i, (s, e, name, q, grid) in enumerate(rows):
        if not pat.search(name):
            continue
        d = e - s
        beside = {}
        j = i - 1
        while j >= 0 and rows[j][0] > s - 50_000_000:
            s2, e2, n2, q2, _ = rows[j]
            if q2 != q and e2 > s:
                beside[short(n2)] = beside.get(short(n2), 0) + (min(e, e2) - max(s, s2))
            j -= 1
        j = i + 1
        while j < len(rows) and rows[j][0] < e:
            s2, e2, n2, q2, _ = rows[j]
            if q2 != q:
                beside[short(n2)] = beside.get(short(n2), 0) + (min(e, e2) - max(s, s2))
            j += 1
        txt = "; ".join(f"{k} {v / d:.2f}" for k, v in sorted(beside.items(), key=lambda kv: -kv[1])[:3]) or "nothing"
        print(f"| `{short(name)}` | {grid[0]} x {grid[1]} x {grid[2]} | {d / 1e3:.1f} | {txt} |")
        shown += 1
        if shown >= limit:
            break


if __name__ == "__main__":
    main()
