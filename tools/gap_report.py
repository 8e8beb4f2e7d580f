#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (largest first), over the last
`frac` of the trace (the timed steps).  Usage: gap_report.py <dir-or-csv> [frac]"""
import csv, glob, os, sys

def main():
    p = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    if os.path.isdir(p):
        p = glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
    rows.sort()
    marker = os.environ.get("GAP_MARKER", "k_read_lengths")
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(starts) >= 2:  # exactly one step: from the second-to-last marker kernel to the last
        rows = rows[starts[-2]:starts[-1]]
    else:
        rows = rows[int(len(rows) * (1 - frac)):]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    print(f"kernels {len(rows)} span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms idle {(span-busy)/1e6:.2f} ms")
    by_name = {}
    for st, en, nm in rows:
        c = by_name.setdefault(nm, [0, 0])
        c[0] += en - st; c[1] += 1
    print("busy time by kernel:")
    for nm, (t, c) in sorted(by_name.items(), key=lambda kv: -kv[1][0])[:45]:
        print(f"  {t/1e6:8.3f} ms  x{c:<4d} {nm}")
    gaps = []
    for a, b in zip(rows, rows[1:]):
        gaps.append((b[0] - a[1], a[2], b[2]))
    by_pair = {}
    for g, x, y in gaps:
        k = (x, y)
        c = by_pair.setdefault(k, [0, 0])
        c[0] += g; c[1] += 1
    top = sorted(gaps, key=lambda g: -g[0])[:12]
    for g, x, y in top:
        print(f"  single gap {g/1e3:9.1f} us  {x} -> {y}")
    print(f"gaps > 20 us: {sum(1 for g in gaps if g[0] > 20000)} totalling {sum(g[0] for g in gaps if g[0] > 20000)/1e6:.2f} ms; "
          f"gaps <= 20 us: {sum(g[0] for g in gaps if g[0] <= 20000)/1e6:.2f} ms")
    for (x, y), (g, c) in sorted(by_pair.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"{g/1e3:9.1f} us  x{c:<4d} {x} -> {y}")

main()
