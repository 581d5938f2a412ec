"""ONE table for the final code -- kernel family -> launches, ms per step, algorithmic flop / bytes, achieved
rate, fraction of the MFMA / HBM peak, bound, counter traffic over algorithmic bytes -- from the round's committed evidence:
    profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace of `bench.py --steps 6 --warmup 2` (tools/kstats.py: us per step)
    profiles/<tag>_bench.json         the bench line (families: in-library timers, algorithmic flop / bytes as the entry points were called)
    profiles/<tag>_pmc.json           three separate rocprofv3 --pmc passes (tools/pmc_summary.py)
usage: python tools/evidence_table.py r05"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
MFMA, HBM = 2500.0, 8000.0   # TFLOP/s dense bf16, GB/s


def family(name):
    n = name
    big = re.search(r"big_nt_kernel<(\d+); (\d+); (\d+); (\d+)>", n)
    if big:
        amode, mode = int(big.group(2)), int(big.group(4))
        if amode in (1, 2):
            return "front-end Conv2d"
        return "CTC head + loss" if mode == 3 else "products fwd / dgrad"
    if "gemm_nt_kernel" in n:
        m = re.search(r"gemm_nt_kernelI\w+?Li(\d+)ELi(\d+)ELi(\d)E", n)
        return "front-end Conv2d" if m and m.group(3) in ("1", "2") else "products fwd / dgrad"
    if "gemm_tn_grouped" in n:
        return "weight gradients"
    if "gemm_tn_kernel" in n:
        m = re.search(r"gemm_tn_kernelI\w+?Li(\d+)ELi(\d+)ELi(\d)E", n)
        return "front-end Conv2d" if m and m.group(3) == "1" else "weight gradients"
    if "attn_bwd_q" in n or "attn_bwd_kv" in n:
        return "attention bwd (kv + q pass)"
    if "attn_bwd_dpos" in n or "attn_dropmask" in n or "attn_bwd_prep" in n or "attn_cast" in n:
        return "attention bwd aux (side stream)"
    if "attn_fwd" in n:
        return "attention fwd"
    if "ln_" in n:
        return "LayerNorm"
    if n.startswith("cf_") or "cf_" in n or "bn_" in n or "dwconv" in n:
        return "convolution module"
    if "conv1_" in n or "specaug" in n:
        return "front-end Conv2d"
    if "ctc_" in n or "lse_fold" in n or "row_lse" in n:
        return "CTC head + loss"
    if "adam" in n or "sqnorm" in n:
        return "optimizer"
    return "other (copies, casts, element-wise)"


rows, US = {}, {}
with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")) as f:
    for r in csv.DictReader(f):
        fam = family(r["kernel"])
        a = rows.setdefault(fam, {"us": 0.0, "calls": 0.0, "kern": []})
        a["us"] += float(r["us_per_step"])
        steps = float(r["total_us"]) / float(r["us_per_step"]) if float(r["us_per_step"]) > 0 else 8.0
        a["calls"] += float(r["calls"]) / steps
        a["kern"].append(r["kernel"])
        US[r["kernel"]] = float(r["us_per_step"])
bench = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")))
pmc = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json")))
fams = bench["families"]
# which timed family of the bench line carries the algorithmic work of which table row
ALG = {"products fwd / dgrad": ["gemm_nt_nn"], "weight gradients": ["gemm_tn"], "attention bwd (kv + q pass)": ["attn_bwd_fused_kernel"],
       "attention fwd": ["attn_fwd_kernel"], "LayerNorm": ["layernorm"], "convolution module": ["conv_module"]}
nsteps = 8.0


AGG = {"gemm_nt_nn", "gemm_tn", "conv_module", "layernorm"}   # family aggregates of pmc_summary.py: not kernels


def pmc_entry(name):
    """the counter summary's entry of one kernel-table name (the two tables spell template arguments differently)"""
    ks = pmc["kernels"]
    if "big_nt_kernel<" in name:
        return ks.get(name.replace("; ", ", ").split("(")[0])
    m = re.search(r"gemm_nt_kernelI\w+?Li(\d+)ELi(\d+)ELi(\d)ELb(\d)", name)
    if m and m.group(3) == "0":
        return ks.get(("gemm_nn" if m.group(4) == "1" else "gemm_nt") + f"[{m.group(1)}x{m.group(2)}]")
    best = None
    for k in ks:
        if k in AGG:
            continue
        core = k.replace("void ", "").split("<")[0]
        if core and core in name and (best is None or len(core) > len(best[0])):
            best = (core, ks[k])
    return best[1] if best else None


def traffic(kernels):
    tot = 0.0
    for k in kernels:
        e = pmc_entry(k)
        if e:
            tot += e["traffic_bytes"] * e["launches"] / nsteps
    return tot


def mfma_util(kernels, us_by_kernel):
    num = den = 0.0
    for k in kernels:
        e = pmc_entry(k)
        if e:
            num += e.get("mfma_util", 0.0) * us_by_kernel[k]
            den += us_by_kernel[k]
    return num / den if den else 0.0


total = sum(a["us"] for a in rows.values())
print(f"`{tag}` evidence, commit `{pmc.get('commit')}`, one box: **{bench['value'] / 1e6:.2f} M frames/s, {bench['ms_per_step']:.1f} ms per optimizer step**"
      f" ({total / 1e3:.1f} ms of kernels under the profiler; side-stream kernels overlap the chain).  Fractions against 2.5 PFLOP/s dense bf16"
      " and 8 TB/s.\n")
print("| family | launches / step | ms / step | algorithmic GFLOP, GB / step | achieved (algorithmic) | of MFMA peak | of HBM peak | MFMA busy (counters) | bound | counter traffic GB / step (x algorithmic) |")
print("|---|---|---|---|---|---|---|---|---|---|")
ANA = {"front-end Conv2d": "front_end_conv2d", "CTC head + loss": "ctc_head_loss",
       "attention bwd aux (side stream)": "attention_aux_side_stream", "optimizer": "optimizer"}
fana = bench.get("families_analytic", {})
for fam, a in sorted(rows.items(), key=lambda kv: -kv[1]["us"]):
    ms = a["us"] / 1e3
    gf = gb = 0.0
    for t in ALG.get(fam, []):
        if t in fams:
            gf += fams[t].get("gflop", 0.0)
            gb += fams[t].get("gbytes", 0.0)
    if fam in ANA and ANA[fam] in fana:   # no in-library counters: the analytic work of bench.py analytic_families
        gf, gb = fana[ANA[fam]]["gflop"], fana[ANA[fam]]["gbytes"]
    tr = traffic(a["kern"]) / 1e9
    mu = mfma_util(a["kern"], US)
    mus = f"{mu:.3f}" if mu > 0 else "—"
    if gf or gb:
        tf, gbs = gf / ms, gb / ms * 1e3
        ai = gf / gb if gb else float("inf")
        bound = "MFMA / latency" if ai > 312 else "HBM / latency"
        alg = f"{gf:.0f}, {gb:.1f}" if gb else f"{gf:.0f}, —"
        ach = (f"{tf:.0f} TFLOP/s" if gf else "") + (", " if gf and gb else "") + (f"{gbs:.0f} GB/s" if gb else "")
        trs = f"{tr:.1f}" + (f" ({tr / gb:.2f} x)" if gb else "")
        print(f"| {fam} | {a['calls']:.0f} | {ms:.2f} | {alg} | {ach} | " + (f"{tf / MFMA:.3f}" if gf else "—") + " | " + (f"{gbs / HBM:.3f}" if gb else "—") + f" | {mus} | {bound} | {trs} |")
    else:
        gbs = tr / ms * 1e3 if tr else 0.0
        print(f"| {fam} | {a['calls']:.0f} | {ms:.2f} | — | {gbs:.0f} GB/s of counter traffic | — | {gbs / HBM:.3f} (traffic) | {mus} | — | {tr:.1f} |")
print(f"| **sum** | {sum(a['calls'] for a in rows.values()):.0f} | {total / 1e3:.2f} | | | | | | | {pmc['total_bytes_per_step'] / 1e9:.1f} (all kernels) |")
