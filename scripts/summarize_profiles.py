#!/usr/bin/env python3
"""Condenses the rocprofv3 output of scripts/gpu_measure.sh <tag> (under gpurun_out/) into the small tracked summaries
under profiles/:

  profiles/<tag>_bench_stdout.json                 the bench line of the plain run
  profiles/<tag>_bench_kernel_stats.csv            rocprofv3 --kernel-trace --stats summary (copied as is)
  profiles/<tag>_kernel_trace_per_launch_shape.csv per (kernel, grid) launch shape: calls, avg/min/max ns, VGPRs, LDS
  profiles/<tag>_pmc_hbm_traffic_per_launch_KB.json  FETCH_SIZE / WRITE_SIZE (KB, raw counter values, summed over the
                                                   XCD rows of a dispatch and averaged over dispatches) per launch shape
  profiles/pmc_hbm_traffic_per_launch_KB.json      copy of the latest one; bench.py reads this file for roofline.traffic

usage: python3 scripts/summarize_profiles.py <tag> [picture passes of the PMC run = 6: 1 warmup + 2 steps + the 3 passes
       of bench.py's per-stage timing]
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    if "at::native" in name:          # torch's own fills / index kernels of the glue
        return "native"
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else "native"


def main():
    tag = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    go, pr = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    os.makedirs(pr, exist_ok=True)
    src = os.path.join(go, f"bench_{tag}.json")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(pr, f"{tag}_bench_stdout.json"))
    st = os.path.join(go, f"prof_{tag}", "bench_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(pr, f"{tag}_bench_kernel_stats.csv"))
    tr = os.path.join(go, f"prof_{tag}", "bench_kernel_trace.csv")
    if os.path.exists(tr):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(tr)):
            k = (short(r["Kernel_Name"]), r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])
            agg.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        with open(os.path.join(pr, f"{tag}_kernel_trace_per_launch_shape.csv"), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["kernel", "grid_x", "workgroup_x", "lds_bytes", "vgpr", "calls", "avg_ns", "min_ns", "max_ns"])
            for k, v in agg.items():
                wr.writerow(list(k) + [len(v), sum(v) // len(v), min(v), max(v)])
    pmc = collections.OrderedDict()
    for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(go, f"pmc_{tag}_{cnt}", "bench_counter_collection.csv")
        if not os.path.exists(p):
            continue
        per_dispatch = collections.OrderedDict()
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] != cnt:
                continue
            k = (f"{short(r['Kernel_Name'])}|grid={r['Grid_Size']}", r["Dispatch_Id"])
            per_dispatch[k] = per_dispatch.get(k, 0.0) + float(r["Counter_Value"])
        shape = collections.OrderedDict()
        for (k, _), v in per_dispatch.items():
            shape.setdefault(k, []).append(v)
        for k, v in shape.items():
            e = pmc.setdefault(k, {})
            e[cnt] = sum(v) / len(v)
            e["launches_per_step"] = len(v) // steps
    # dynamic instruction counters per launch shape (the valu_issue roofline of bench.py prices SQ_INSTS_VALU)
    pi = os.path.join(go, f"pmc_{tag}_INSTS", "bench_counter_collection.csv")
    if os.path.exists(pi):
        per_dispatch = collections.OrderedDict()
        for r in csv.DictReader(open(pi)):
            k = (f"{short(r['Kernel_Name'])}|grid={r['Grid_Size']}", r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[k] = per_dispatch.get(k, 0.0) + float(r["Counter_Value"])
        shape = collections.OrderedDict()
        for (k, _, cnt), v in per_dispatch.items():
            shape.setdefault(k, {}).setdefault(cnt, []).append(v)
        insts = collections.OrderedDict()
        for k, d in shape.items():
            if k.startswith("native"):
                continue
            if len(next(iter(d.values()))) < steps:      # not a launch of the step (the bench's one-off statistic passes, e.g. FrameHotPath.sad_candidates' table-driven searches)
                continue
            e = {cnt: sum(v) / len(v) for cnt, v in d.items()}
            e["launches_per_step"] = len(next(iter(d.values()))) // steps
            if e.get("SQ_WAVES"):
                e["valu_per_wave"] = e.get("SQ_INSTS_VALU", 0) / e["SQ_WAVES"]
                e["salu_per_wave"] = e.get("SQ_INSTS_SALU", 0) / e["SQ_WAVES"]
            insts[k] = e
        # the bench line of the PMC run itself names the command and the kernel sources the counters belong to: bench.py forms a roofline fraction only
        # from counters whose key equals its own (stale counters must not price another workload)
        try:
            meta = json.loads(open(os.path.join(go, f"pmc_{tag}_INSTS", "stdout.json")).read().strip().splitlines()[-1]).get("workload_key")
        except Exception:
            meta = None
        if meta:
            insts["_meta"] = meta
        for name in (f"{tag}_pmc_insts_per_launch.json", "pmc_insts_per_launch.json"):
            json.dump(insts, open(os.path.join(pr, name), "w"), indent=1)
    if pmc:
        for name in (f"{tag}_pmc_hbm_traffic_per_launch_KB.json", "pmc_hbm_traffic_per_launch_KB.json"):
            json.dump(pmc, open(os.path.join(pr, name), "w"), indent=1)
    print("profiles written for", tag)


if __name__ == "__main__":
    main()
