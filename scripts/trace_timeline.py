#!/usr/bin/env python3
"""Timeline summary of ONE picture out of a rocprofv3 kernel trace (scripts/gpu_trace.sh): per hardware queue the busy time, the span and
the largest kernels; the picture = the last run of dispatches between two `pis_cands_kernel` launches of the first level.

usage: scripts/trace_timeline.py <dir with *_kernel_trace.csv> [picture index from the end, default 1]"""
import csv, glob, sys, collections, re

def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)[:60]

def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f = glob.glob(d + "/*kernel_trace.csv")[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a picture starts with the stage-0 candidate kernel of the first (largest-PU) level: the dispatches with the smallest grid of that kernel
    starts = [i for i, r in enumerate(rows) if "pis_cands_kernel" in r["Kernel_Name"]]
    if not starts:
        print("no pis_cands_kernel in the trace"); return
    g0 = min(int(rows[i]["Grid_Size_X"]) for i in starts)
    first = [i for i in starts if int(rows[i]["Grid_Size_X"]) == g0]
    if len(first) < back + 1:
        print("not enough pictures in the trace"); return
    a, b = first[-back - 1], first[-back]
    pic = rows[a:b]
    t0 = int(pic[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in pic)
    print("picture: %d dispatches, span %.1f us" % (len(pic), (t1 - t0) / 1e3))
    byq = collections.defaultdict(list)
    for r in pic: byq[r["Queue_Id"]].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
        s0 = min(int(r["Start_Timestamp"]) for r in rs); s1 = max(int(r["End_Timestamp"]) for r in rs)
        print("queue %s: %4d dispatches, busy %8.1f us, first start +%.1f us, last end +%.1f us" % (q, len(rs), busy / 1e3, (s0 - t0) / 1e3, (s1 - t0) / 1e3))
        k = collections.Counter()
        for r in rs: k[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for name, ns in k.most_common(6):
            print("    %-60s %8.1f us" % (name, ns / 1e3))
    # union of busy intervals over all queues = time at least one kernel was running
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in pic)
    u = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    u += ce - cs
    print("at least one kernel running: %.1f us of %.1f us" % (u / 1e3, (t1 - t0) / 1e3))

if __name__ == "__main__":
    main()
