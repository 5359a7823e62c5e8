#!/usr/bin/env python3
"""Average SQ / GRBM counters per kernel from one or more rocprofv3 `--pmc` counter_collection.csv files (separate
passes may be given together) -> a markdown table + derived figures, for profiles/.

    python scripts/summarize_sq.py pass1.csv [pass2.csv ...] --match rank_match --elements 134217728 --out profiles/x.md

--elements: keys (or fp32 elements) one dispatch of the matched kernel processes; adds "wave-instructions per 64 elements".
Counter units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_BUSY_CYCLES per SE; SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs; SQ_INSTS_* wave-instructions; GRBM_GUI_ACTIVE
shader-clock cycles of the dispatch."""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csvs", nargs="+")
    ap.add_argument("--match", default="", help="only kernels whose name contains this")
    ap.add_argument("--elements", type=float, default=0.0)
    ap.add_argument("--skip", type=int, default=0, help="ignore the first N dispatches of each kernel (warm-up)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--title", default="")
    ap.add_argument("--command", default="")
    args = ap.parse_args()
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    for path in args.csvs:
        rows = list(csv.DictReader(open(path)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        seen = collections.defaultdict(set)
        for r in rows:
            k = short(r["Kernel_Name"])
            if args.match and args.match not in k:
                continue
            disp = r.get("Dispatch_Id", r["Start_Timestamp"])
            seen[k].add(disp)
            if len(seen[k]) <= args.skip:
                continue
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "End_Timestamp" in r and r["Counter_Name"] == rows[0]["Counter_Name"]:
                durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = [f"# {args.title or 'SQ / GRBM counters per kernel'}", ""]
    if args.command:
        lines += [f"command: `{args.command}`", ""]
    lines += ["Averages over the dispatches of each kernel (rocprofv3 --pmc, kernel-trace only; profiled passes run at lower "
              "clocks than un-profiled ones, so durations here are NOT the bench's).", ""]
    for k, cs in vals.items():
        lines += [f"## `{k}`", "", "| counter | mean per dispatch | dispatches |", "|---|---:|---:|"]
        mean = {}
        for c, v in sorted(cs.items()):
            mean[c] = sum(v) / len(v)
            lines.append(f"| {c} | {mean[c]:.4g} | {len(v)} |")
        if durs[k]:
            lines.append(f"| duration (us, profiled) | {sum(durs[k]) / len(durs[k]):.1f} | {len(durs[k])} |")
        lines.append("")
        d = []
        insts = [mean.get(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                                       "SQ_INSTS_SMEM")]
        if args.elements and any(i is not None for i in insts):
            tot = sum(i for i in insts if i is not None)
            d.append(f"wave-instructions per 64 elements (VALU + SALU + LDS [+ VMEM + SMEM] counted): **{tot * 64 / args.elements:.1f}** "
                     f"(VALU {64 * (mean.get('SQ_INSTS_VALU') or 0) / args.elements:.1f}, SALU {64 * (mean.get('SQ_INSTS_SALU') or 0) / args.elements:.1f}, "
                     f"LDS {64 * (mean.get('SQ_INSTS_LDS') or 0) / args.elements:.1f})")
        if "SQ_WAVE_CYCLES" in mean:
            wc = mean["SQ_WAVE_CYCLES"]
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"):
                if c in mean:
                    d.append(f"{c} / SQ_WAVE_CYCLES = {mean[c] / wc:.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (one GRBM each): / 8 = shader cycles of the dispatch
            d.append(f"MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = "
                     f"**{mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (mean['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}**")
        if "GRBM_GUI_ACTIVE" in mean and durs[k]:
            d.append(f"effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / duration = {mean['GRBM_GUI_ACTIVE'] / 8 / (sum(durs[k]) / len(durs[k])) / 1e3:.2f} GHz")
        lines += [f"* {x}" for x in d] + [""]
    open(args.out, "w").write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
