#!/usr/bin/env python3
"""One timed step of a rocprofv3 --kernel-trace CSV as a timeline: runs of the same kernel collapsed, with the busy
time of the run and the idle time before it.  Usage: step_timeline.py <dir-or-csv>"""
import csv, glob, os, sys

def main():
    p = sys.argv[1]
    if os.path.isdir(p):
        p = glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]))
    rows.sort()
    marker = os.environ.get("GAP_MARKER", "k_read_lengths")
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    rows = rows[starts[-2]:starts[-1]]
    t0 = rows[0][0]
    prev_end = rows[0][0]
    out = []
    for st, en, nm in rows:
        gap = st - prev_end
        if out and out[-1][0] == nm and gap < 20000:
            out[-1][1] += en - st; out[-1][2] += gap; out[-1][3] += 1
        else:
            out.append([nm, en - st, gap, 1, st - t0])
        prev_end = en
    for nm, busy, gap, c, at in out:
        print(f"{at/1e6:8.3f} ms  idle-before {gap/1e3:7.1f} us  busy {busy/1e3:8.1f} us  x{c:<3d} {nm}")

main()
