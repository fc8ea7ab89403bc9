#!/usr/bin/env python3
"""Condenses gpurun_out/probe_<tag>/pass*/p_counter_collection.csv (scripts/gpu_pmc_probe.sh) into one line pair per launch shape.
usage: python3 scripts/probe_report.py <tag> [> profiles/<round>_probe_<tag>.txt]"""
import collections
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "probe_" + tag, "pass*", "p_counter_collection.csv"))):
        seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
            k = (m.group(1) if m else r["Kernel_Name"][:40], r["Grid_Size"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
        for k, v in seen.items():
            disp[k] = len(v)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
        n = disp[k]
        g = lambda c: v.get(c, 0) / n   # noqa: E731
        wc = g("SQ_WAVE_CYCLES")
        if not wc:
            continue
        print("%s grid %s (%d dispatches)" % (k[0], k[1], n))
        print("   wave-cycles %.3g | of those: waiting on a dependency %.2f (on LDS %.2f) | issuing VALU %.2f  SALU %.2f  LDS %.2f  VMEM %.2f"
              % (wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_SCA") / wc, g("SQ_ACTIVE_INST_LDS") / wc,
                 g("SQ_ACTIVE_INST_VMEM") / wc))
        print("   per wave: VALU %.0f  SALU %.0f  LDS %.0f  VMEM-read %.0f instructions | LDS bank-conflict cycles / LDS instruction %.2f | waves %.0f"
              % (g("SQ_INSTS_VALU") / max(1, g("SQ_WAVES")), g("SQ_INSTS_SALU") / max(1, g("SQ_WAVES")), g("SQ_INSTS_LDS") / max(1, g("SQ_WAVES")),
                 g("SQ_INSTS_VMEM_RD") / max(1, g("SQ_WAVES")), g("SQ_LDS_BANK_CONFLICT") / max(1, g("SQ_INSTS_LDS")), g("SQ_WAVES")))


if __name__ == "__main__":
    main()
