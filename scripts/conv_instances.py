#!/usr/bin/env python3
"""The longest individual kernel instances of a rocprofv3 kernel trace, grouped by (kernel, grid): which convolution launches
cost what at a given batch size.   python scripts/conv_instances.py <kernel_trace.csv> [top]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"]
    if name.startswith("optex::") or "optex" in name[:20]:
        continue
    key = (name[:70], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"), r.get("Workgroup_Size_X", "?"))
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1
    agg[key][1] += us
tot = sum(v[1] for v in agg.values())
print(f"non-optex kernel time {tot / 1e3:.2f} ms over the whole trace; top {top} (kernel, grid x/y/z, workgroup) by total time")
print("| kernel | grid | wg | calls | total ms | avg us |\n|---|---|---:|---:|---:|---:|")
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"| `{key[0]}` | {key[1]}x{key[2]}x{key[3]} | {key[4]} | {n} | {us / 1e3:.3f} | {us / n:.1f} |")
