"""Summarises a rocprofv3 --kernel-trace of CU-level calls (one vtmhip_predInterSearch_batch_dev per CU): launches per call, busy time, gaps.
A call = the run of launches between two host synchronisations; it starts with the AMVP prediction kernel (motion_comp_amvp_kernel)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)]
rows = []
for p in f:
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]))
rows.sort()
calls, cur = [], []
for r in rows:
    if "amvp" in r[2] and "select" not in r[2] and cur:
        calls.append(cur); cur = []
    cur.append(r)
if cur: calls.append(cur)
calls = [c for c in calls if len(c) >= 5]
n = len(calls)
print("calls", n, "launches", sum(len(c) for c in calls))
lens = collections.Counter(len(c) for c in calls)
print("launches per call:", sorted(lens.items()))
busy = sum(sum(e - s for s, e, _ in c) for c in calls) / n / 1e3
span = sum(c[-1][1] - c[0][0] for c in calls) / n / 1e3
print("per call: kernel busy %.1f us, first start -> last end %.1f us, gaps %.1f us" % (busy, span, span - busy))
gaps = []
for c in calls:
    for a, b in zip(c, c[1:]):
        gaps.append((b[0] - a[1]) / 1e3)
gaps.sort()
print("gap between consecutive launches of a call: median %.2f us, p90 %.2f us, mean %.2f us" % (gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)], sum(gaps) / len(gaps)))
per = collections.defaultdict(lambda: [0, 0.0])
for c in calls:
    for s, e, k in c:
        per[k][0] += 1; per[k][1] += (e - s) / 1e3
print("%-60s %8s %10s %10s" % ("kernel", "per call", "us each", "us / call"))
for k, (cnt, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %8.2f %10.2f %10.2f" % (k[:60], cnt / n, us / cnt, us / n))
# between calls: last end of call i -> first start of call i + 1 (download, host glue of the test, upload)
inter = sorted((b[0][0] - a[-1][1]) / 1e3 for a, b in zip(calls, calls[1:]))
print("between calls (download + host + upload): median %.1f us" % inter[len(inter) // 2])
