#!/usr/bin/env python3
"""One picture of an OVERLAPPED run's rocprofv3 kernel trace in time slices: per slice, which kernels run (share of the slice each is active) -- where the machine runs one
small kernel at a time and where the streams overlap.  usage: scripts/timeline_slices.py <dir with *_kernel_trace.csv> [slice us = 250]"""
import csv, glob, sys, collections, re

def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"[(<].*$", "", name)[:28]

d = sys.argv[1]; sl = float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
rows = [r for r in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "pis_cands_kernel" in r["Kernel_Name"]]
g0 = min(int(rows[i]["Grid_Size_X"]) for i in starts)
first = [i for i in starts if int(rows[i]["Grid_Size_X"]) == g0]
a, b = first[-2], first[-1]
pic = rows[a:b]
t0 = int(pic[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in pic)
print("picture: %d dispatches, span %.1f us" % (len(pic), (t1 - t0) / 1e3))
n = int((t1 - t0) / 1e3 / sl) + 1
acc = [collections.defaultdict(float) for _ in range(n)]
for r in pic:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    k = short(r["Kernel_Name"]) + ":" + r["Grid_Size_X"]
    i = int(s / sl)
    while i * sl < e and i < n:
        lo, hi = max(s, i * sl), min(e, (i + 1) * sl)
        if hi > lo: acc[i][k] += (hi - lo) / sl
        i += 1
for i, a_ in enumerate(acc):
    tot = sum(a_.values())
    print("%6.0f us  conc %.2f  %s" % (i * sl, tot, "  ".join("%s %.2f" % (k, v) for k, v in sorted(a_.items(), key=lambda x: -x[1])[:6])))
