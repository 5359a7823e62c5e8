#!/usr/bin/env python3
"""Turn the raw slabs of scripts/gemm_timeline_probe.bin into a markdown timeline (profiles/r05_gemm_timeline.md).

    python scripts/gemm_timeline_report.py out_rs.bin [out_lds.bin ...] > timeline.md

Units: shader cycles (s_memtime); 64 MFMAs (v_mfma_f32_16x16x4_f32, 32 cycles per SIMD each) = 2048 cycles of matrix pipe."""
import sys

import numpy as np


def load(path):
    raw = np.fromfile(path, dtype=np.uint64)
    magic, kind, waves, words, S, n, us1000, reps = [int(v) for v in raw[:8].astype(np.int64)]
    assert magic == 0x4C54474D, "not a timeline file"
    return kind, S, n, us1000 / 1000.0, raw[8:].reshape(waves, words).astype(np.int64)


def pct(a, q):
    return float(np.percentile(a, q)) if len(a) else float("nan")


def row(name, a, ideal=None):
    a = np.asarray(a, dtype=np.float64)
    s = f"| {name} | {a.mean():8.0f} | {pct(a, 50):8.0f} | {pct(a, 10):8.0f} | {pct(a, 90):8.0f} | {a.max():8.0f} |"
    if ideal:
        s += f" {ideal / a.mean():.3f} |"
    else:
        s += " |"
    return s


HEAD = "| interval | mean | median | p10 | p90 | max | MFMA pipe busy (ideal / mean) |\n|---|---:|---:|---:|---:|---:|---:|"


def report_rs(path):
    kind, S, n, us, d = load(path)
    tiles = d[:, 0] >> 8
    xcc = d[:, 0] & 0xFF
    T = int(tiles.max())
    print(f"\n## R-stationary `gemm_rs_kernel<4, 64>`: [{S}, 256, {n}] x 256^2, {us:.1f} us per launch (stamped build), "
          f"{len(d)} wavefronts, {int(tiles.min())}..{T} tiles per workgroup\n")
    real0, entry, issued = d[:, 1], d[:, 2], d[:, 3]
    per = d[:, 4:4 + T * 18].reshape(len(d), T, 18)
    # real time at the end sits behind the wave's last tile
    real1 = np.array([d[w, 4 + int(tiles[w]) * 18] for w in range(len(d))])
    last = np.array([per[w, int(tiles[w]) - 1, 17] for w in range(len(d))])
    cyc, ns = last - entry, (real1 - real0) * 10.0
    print(f"effective shader clock over the kernel (cycles / 100 MHz real time), median over wavefronts: "
          f"**{np.median(cyc / ns):.3f} GHz**; wavefront lifetime median {np.median(ns) / 1e3:.1f} us "
          f"(p10 {pct(ns, 10) / 1e3:.1f}, p90 {pct(ns, 90) / 1e3:.1f})\n")
    print("Prologue (cycles):\n")
    print(HEAD)
    print(row("entry -> matrix (256 loads) + B ring (16 loads) issued", issued - entry))
    print(row("issued -> first tile's first 64 MFMAs issued (waits for the matrix)", per[:, 0, 1] - issued))
    print()
    ok = np.arange(T)[None, :] < tiles[:, None]
    steady = ok.copy()
    steady[:, 0] = False
    k = np.diff(per[:, :, :17], axis=2)          # 16 intervals of 4 k-steps
    epi = per[:, :, 17] - per[:, :, 16]           # stores issued
    nxt = per[:, 1:, 0] - per[:, :-1, 17]         # loop turn
    whole = per[:, :, 17] - per[:, :, 0]
    print("Per tile, steady state (tiles 1.. of every wavefront), cycles; 4 k-steps = 64 MFMAs = 2048 cycles of pipe:\n")
    print(HEAD)
    for i in range(16):
        print(row(f"k-steps {4 * i}..{4 * i + 3}", k[:, :, i][steady], 2048))
    print(row("epilogue (accumulators -> 16 x 16-byte stores per lane issued)", epi[steady]))
    print(row("loop turn (next tile's pointers)", nxt[ok[:, 1:]]))
    print(row("whole tile", whole[steady], 16 * 2048))
    print()
    print("First tile of a wavefront:\n")
    print(HEAD)
    for i in (0, 1, 2, 3, 15):
        print(row(f"k-steps {4 * i}..{4 * i + 3}", k[:, 0, i], 2048))
    print(row("whole tile", whole[:, 0], 16 * 2048))
    print()
    mf = k[steady.nonzero()[0], steady.nonzero()[1], :].sum(axis=1)
    tot = whole[steady]
    print(f"share of a steady tile spent in the k-loop: {mf.sum() / tot.sum():.3f}; in the epilogue: {epi[steady].sum() / tot.sum():.3f}")
    slow = k[steady] > 1.15 * 2048
    print(f"4-k-step intervals more than 15 % over the pipe time: {slow.mean() * 100:.1f} % of all, "
          f"they hold {((k[steady] - 2048) * slow).sum() / max(1, (k[steady] - 2048).clip(0).sum()) * 100:.0f} % of the excess cycles\n")
    print("By XCD (whole steady tile, mean cycles): " + ", ".join(
        f"{x}: {whole[(xcc == x)][steady[xcc == x]].mean():.0f}" for x in sorted(set(xcc.tolist()))))


def report_lds(path):
    kind, S, n, us, d = load(path)
    print(f"\n## LDS-tiled `gemm16_cm_kernel<256, 128, 16, 4, 2>`: [{S}, 256, {n}] x 256^2, {us:.1f} us per launch (stamped build), "
          f"{len(d)} wavefronts sampled (8 per 256 x 128 tile, two per SIMD)\n")
    real0, entry, staged = d[:, 1], d[:, 2], d[:, 3]
    ch = d[:, 4:4 + 64].reshape(len(d), 16, 4)
    stored, real1 = d[:, 68], d[:, 69]
    cyc, ns = stored - entry, (real1 - real0) * 10.0
    print(f"effective shader clock: **{np.median(cyc / ns):.3f} GHz**; tile lifetime median {np.median(ns) / 1e3:.2f} us\n")
    start = np.concatenate([staged[:, None], ch[:, :-1, 3]], axis=1)   # chunk start = previous barrier passed
    print("Per 16-deep K chunk (a wavefront issues 64 MFMAs = 2048 cycles; the two wavefronts of a SIMD share its pipe: "
          "4096 cycles of pipe per chunk), cycles:\n")
    print(HEAD)
    print(row("entry -> first chunk staged (global -> LDS, barrier)", staged - entry))
    print(row("chunk start -> next chunk's global loads issued", (ch[:, :, 0] - start)[:, 1:15]))
    print(row("-> 64 MFMAs issued (LDS fragment reads inside)", (ch[:, :, 1] - ch[:, :, 0])[:, 1:15], 2048))
    print(row("-> LDS refilled (waits for the global loads)", (ch[:, :, 2] - ch[:, :, 1])[:, 1:15]))
    print(row("-> barrier passed", (ch[:, :, 3] - ch[:, :, 2])[:, 1:15]))
    print(row("whole chunk (chunks 1..14)", (ch[:, :, 3] - start)[:, 1:15], 4096))
    print(row("last chunk -> stores issued (epilogue)", stored - ch[:, 15, 3]))
    print(row("whole tile", stored - entry, 16 * 4096))
    print()
    print("Whole chunk by chunk index (mean cycles): " + " ".join(f"{v:.0f}" for v in (ch[:, :, 3] - start).mean(axis=0)))


if __name__ == "__main__":
    print("# s_memtime timeline of the rotation GEMMs (scripts/gemm_timeline_probe.hip, -DOPTEX_TIMELINE)")
    for p in sys.argv[1:]:
        kind = int(np.fromfile(p, dtype=np.int64, count=2)[1])
        (report_rs if kind == 0 else report_lds)(p)
