#!/usr/bin/env python3
"""Prices the vector-ALU instruction mix of every kernel in vtm_amd/libvtmhip.so with the MEASURED gfx950 issue costs
(scripts/valu_issue.hip -> profiles/r02_valu_issue.jsonl), for the `valu_issue` roofline of bench.py / DESIGN.md section 4.

    python3 scripts/isa_mix.py [--valu profiles/r02_valu_issue.jsonl] [--out profiles/isa_mix.json]

Method: the code objects are extracted from the shared object (llvm-objdump --offloading) and disassembled; inside each kernel every
backward branch closes a loop, an instruction's weight is LOOP_WEIGHT ** (number of loops around it) -- the usual static profile -- which
gives the kernel's share of HALF-RATE opcodes (per-opcode cost > 3 cycles in the micro-benchmark; unmeasured opcodes count as full rate).
`cycles_per_valu_inst` is then read off the measured MIXED-stream table (the "imix" rows of the micro-benchmark: 0 % half-rate 2.1 cycles,
12.5 % 3.5, 25 % 3.8, 50 % 3.96, 75 %+ 4.0 at 4 waves per SIMD): on gfx950 a stream with any sizeable share of half-rate opcodes issues at
3.5 .. 4.0 cycles per wave64 instruction regardless of their order -- NOT at the share-weighted sum of the two rates (that additive model, used
until round-2 mid-way, put the SATD kernel above its own roof).  `additive_cycles_per_valu_inst` keeps the old figure for comparison.
Also reported per kernel: the static share of half-rate instructions and the VALU : SALU : LDS : VMEM weighted counts."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LOOP_WEIGHT = 16.0
FULL_RATE = 2.11   # measured: v_add_u32 / v_sub_u32 / v_and_b32 / v_ashrrev_i32 / v_mov_b32 with >= 2 waves per SIMD


def load_costs(path):
    """{base opcode: cycles per wave-instruction per SIMD}: max-based steady-state figure at 4 waves per SIMD, 8 independent chains"""
    cost = {}
    for line in open(path):
        r = json.loads(line)
        if r["chains"] != 8 or r["waves_per_simd"] != 4 or r["op"].startswith(("mix:", "pair:", "s_")):
            continue
        if r["op"] == "v_cndmask_b32":   # the row with a never-written vcc as mask measures 22.7 cycles (an artefact of that benchmark body: the same
            continue                     # instruction behind a v_cmp, or with an SGPR-pair mask, costs 4.1): the "(sgpr-pair mask)" row prices it
        name = r["op"].split()[0]
        c = r["cycles_per_inst_per_simd"] * r["max_over_mean"]
        key = name
        if "dpp" in r["op"]:
            key = "dpp"
        elif "sdwa" in r["op"]:
            key = "sdwa"
        cost[key] = max(cost.get(key, 0.0), c)
    return cost


def load_mix_table(path):
    """[(half-rate share, cycles per instruction)] from the independent-chain mixes at 4 waves per SIMD, ascending, made monotone"""
    share = {"eight different full-rate": 0.0, "7 full : 1 half": 0.125, "3 full : 1 half": 0.25, "add_u32 / pk_add_i16 alternating": 0.5, "1 full : 3 half": 0.75,
             "eight different half-rate": 1.0}
    pts = {}
    for line in open(path):
        r = json.loads(line)
        if not r["op"].startswith("imix:") or r["waves_per_simd"] != 4:
            continue
        for key, sh in share.items():
            if key in r["op"]:
                pts[sh] = r["cycles_per_inst_per_simd"] * r["max_over_mean"]
    tab = sorted(pts.items())
    for i in range(1, len(tab)):
        tab[i] = (tab[i][0], max(tab[i][1], tab[i - 1][1]))
    return tab


def interp(tab, x):
    for (x0, y0), (x1, y1) in zip(tab, tab[1:]):
        if x <= x1:
            return y0 + (y1 - y0) * (x - x0) / (x1 - x0)
    return tab[-1][1]


def base_op(mn):
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        if mn.endswith(suf):
            return mn[:-len(suf)], suf
    return mn, ""


def price(mn, cost):
    b, suf = base_op(mn)
    if suf == "_dpp":
        return cost.get("dpp", FULL_RATE), True
    if suf == "_sdwa":
        return cost.get("sdwa", FULL_RATE), True
    if b in cost:
        return cost[b], True
    # families measured through one member
    fam = {"v_pk_": "v_pk_add_i16", "v_cmp_": "v_cmp_lt_i32", "v_cmpx_": "v_cmp_lt_i32", "v_min_": "v_max_i32", "v_max_": "v_max_i32",
           "v_med3": "v_max3_i32", "v_min3": "v_max3_i32", "v_max3": "v_max3_i32", "v_dot": "v_dot2_i32_i16", "v_mad_": "v_mad_i32_i24",
           "v_lshlrev_b64": "v_lshl_add_u64", "v_lshrrev_b64": "v_lshl_add_u64", "v_ashrrev_i64": "v_lshl_add_u64", "v_add_lshl": "v_lshl_add_u32",
           "v_and_or": "v_lshl_or_b32", "v_or3": "v_add3_u32", "v_xad": "v_add3_u32", "v_readfirstlane": "v_readlane_b32", "v_writelane": "v_readlane_b32",
           "v_sub_co": "v_addc_co_u32", "v_add_co": "v_addc_co_u32", "v_subb_co": "v_addc_co_u32", "v_subbrev": "v_addc_co_u32", "v_subrev_co": "v_addc_co_u32",
           "v_lshrrev_b32": "v_lshrrev_b32", "v_sad": "v_sad_u16", "v_bfe": "v_bfe_i32", "v_mul_f64": "v_mul_f64", "v_fma_f64": "v_mul_f64", "v_fmac_f64": "v_mul_f64",
           "v_add_f64": "v_mul_f64", "v_cvt_": "v_cvt_f64_i32"}
    for pre, rep in fam.items():
        if b.startswith(pre) and rep in cost:
            return cost[rep], True
    return FULL_RATE, False


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="isa_mix_")
    so = os.path.join(tmp, "lib.so")
    subprocess.check_call(["cp", lib, so])
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = []
    for f in sorted(os.listdir(tmp)):
        if "gfx950" in f:
            text.append(subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, f)]).decode())
    subprocess.call(["rm", "-rf", tmp])
    return "\n".join(text)


def demangle(names):
    out = subprocess.check_output(["c++filt"], input="\n".join(names).encode()).decode().split("\n")
    return dict(zip(names, out))


def short(name):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "vtm_amd", "libvtmhip.so"))
    ap.add_argument("--valu", default=os.path.join(ROOT, "profiles", "r02_valu_issue.jsonl"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "isa_mix.json"))
    a = ap.parse_args()
    cost = load_costs(a.valu)
    mix_tab = load_mix_table(a.valu)
    assert len(mix_tab) >= 5, "the micro-benchmark file lacks the imix rows"
    txt = disassemble(a.lib)
    funcs = collections.OrderedDict()
    cur = None
    for line in txt.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(\S+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            base = int(m.group(1), 16)
            continue
        if cur is None or not line.startswith("\t"):
            continue
        m = re.match(r"^\t(\S+)\s*(.*?)\s*// ([0-9A-F]+):.*?(?:<\S+\+0x([0-9a-f]+)>)?$", line)
        if not m:
            continue
        mn, addr, tgt = m.group(1), int(m.group(3), 16) - base, m.group(4)
        cur.append((addr, mn, int(tgt, 16) if (tgt and mn.startswith("s_cbranch") or tgt and mn == "s_branch") else None))
    dm = demangle(list(funcs))
    result = collections.OrderedDict()
    unknown = collections.Counter()
    for f, ins in funcs.items():
        if not ins:
            continue
        loops = [(t, a0) for (a0, mn, t) in ins if t is not None and t <= a0]
        tot = collections.defaultdict(float)
        wsum = wcost = whalf = 0.0
        for a0, mn, _ in ins:
            depth = sum(1 for (lo, hi) in loops if lo <= a0 <= hi)
            w = LOOP_WEIGHT ** min(depth, 4)
            if mn.startswith("v_"):
                c, known = price(mn, cost)
                if not known:
                    unknown[base_op(mn)[0]] += 1
                wsum += w
                wcost += w * c
                whalf += w * (c > 3.0)
                tot["valu"] += w
            elif mn.startswith("s_"):
                tot["salu"] += w
            elif mn.startswith("ds_"):
                tot["lds"] += w
            elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
                tot["vmem"] += w
        if wsum == 0:
            continue
        k = short(dm.get(f, f))
        result[k] = {"cycles_per_valu_inst": round(interp(mix_tab, whalf / wsum), 3), "additive_cycles_per_valu_inst": round(wcost / wsum, 3), "half_rate_share": round(whalf / wsum, 3), "static_valu": sum(1 for i in ins if i[1].startswith("v_")),
                     "loops": len(loops), "weighted_mix": {kk: round(v / wsum, 3) for kk, v in tot.items()}}
    meta = {"_method": "static loop-weighted (x%g per loop level) share of half-rate opcodes, priced with the measured mixed-stream issue cost (imix rows, 4 waves per SIMD); "
                       "cycles per wave64 instruction per SIMD; unmeasured opcodes count as full rate" % LOOP_WEIGHT, "_mix_table": mix_tab, "_source": os.path.relpath(a.valu, ROOT), "_unmeasured_opcodes": dict(unknown.most_common(40))}
    meta.update(result)
    json.dump(meta, open(a.out, "w"), indent=1)
    for k, v in result.items():
        print("%-48s %.2f cycles/VALU inst, %2.0f%% half-rate, %d static VALU, %d loops" % (k, v["cycles_per_valu_inst"], 100 * v["half_rate_share"], v["static_valu"], v["loops"]))
    print("unmeasured (priced at full rate):", dict(unknown.most_common(15)), file=sys.stderr)


if __name__ == "__main__":
    main()
