#!/usr/bin/env python3
"""How busy is the GPU in the last WINDOW ms of a rocprofv3 kernel trace (all queues merged), and how large are the idle gaps
between consecutive kernels?  For launch-bound runs (one texture per call): what a graph replay could still return.
    python scripts/busy_fraction.py <kernel_trace.csv> [window_ms = 150]"""
import csv
import sys


def main():
    path = sys.argv[1]
    window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 150e6
    iv = []
    with open(path) as f:
        for r in csv.DictReader(f):
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    iv.sort()
    t_end = max(e for _, e in iv)
    t_lo = t_end - window
    iv = [(max(s, t_lo), e) for s, e in iv if e > t_lo]
    busy, gaps, cur_s, cur_e = 0, [], iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = cur_e - iv[0][0]
    print(f"last {span / 1e6:.1f} ms of the trace: {len(iv)} kernels, busy {busy / 1e6:.1f} ms = {busy / span:.3f}, idle {sum(gaps) / 1e6:.1f} ms in {len(gaps)} gaps")
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)):
        g = [x for x in gaps if lo * 1e3 <= x < hi * 1e3]
        print(f"  gaps of {lo}-{hi if hi < 1e9 else 'inf'} us: {len(g):5d}, {sum(g) / 1e6:6.2f} ms")


if __name__ == "__main__":
    main()
