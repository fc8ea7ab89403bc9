#!/usr/bin/env python3
"""Prints the per-launch-shape table of profiles/<tag>_kernel_trace_per_launch_shape.csv sorted by share, with cycles per VALU instruction and HBM
traffic when the PMC summaries of the same tag (or of --pmc-tag) exist.  usage: python3 scripts/top_kernels.py <tag> [--pmc-tag T] [--top N]"""
import argparse
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--pmc-tag", default=None)
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--steps", type=int, default=15, help="picture passes of the traced run (12 + the 3 of the per-stage timing)")
    a = ap.parse_args()
    pr = os.path.join(ROOT, "profiles")
    rows = list(csv.DictReader(open(os.path.join(pr, a.tag + "_kernel_trace_per_launch_shape.csv"))))

    def load(name):
        try:
            return json.load(open(os.path.join(pr, (a.pmc_tag or a.tag) + name)))
        except OSError:
            return {}
    ins, hb = load("_pmc_insts_per_launch.json"), load("_pmc_hbm_traffic_per_launch_KB.json")
    rows.sort(key=lambda r: -int(r["calls"]) * int(r["avg_ns"]))
    tot = sum(int(r["calls"]) * int(r["avg_ns"]) for r in rows)
    print("kernel time per picture pass: %.3f ms" % (tot / a.steps / 1e6))
    for r in rows[:a.top]:
        key = "%s|grid=%s" % (r["kernel"], r["grid_x"])
        i, h = ins.get(key, {}), hb.get(key, {})
        v, t = i.get("SQ_INSTS_VALU", 0), int(r["avg_ns"]) * 1e-9
        mb = (2 * h.get("FETCH_SIZE", 0) + h.get("WRITE_SIZE", 0)) / 1024
        print("%-32s grid %-9s lds %-6s vgpr %-4s calls %-4s avg_us %8.1f ms/pass %6.3f | cyc/VALU %5.2f valu/wave %7.0f salu/wave %6.0f | HBM MB %7.1f GB/s %6.0f"
              % (r["kernel"][:32], r["grid_x"], r["lds_bytes"], r["vgpr"], r["calls"], int(r["avg_ns"]) / 1e3, int(r["calls"]) * int(r["avg_ns"]) / a.steps / 1e6,
                 1024 * 2.4e9 * t / v if v else 0, i.get("valu_per_wave", 0), i.get("salu_per_wave", 0), mb, mb / 1e3 / t if t else 0))


if __name__ == "__main__":
    main()
