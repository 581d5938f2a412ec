"""Per-kernel summary of three separate rocprofv3 --pmc passes of the same bench command (rocpd databases):
  pass 1  FETCH_SIZE                                 (TCC: 3 of 4 slots)
  pass 2  WRITE_SIZE                                 (TCC: 2 slots)
  pass 3  SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   (SQ + GRBM)
as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-byte
read requests at 64 bytes, so reads are doubled; WRITE_SIZE is exact for 16-byte stores and f32 atomics.  MFMA utilisation of a
kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums
the 8 XCDs).
Besides the per-kernel entries the output holds one aggregated entry per kernel FAMILY of bench.py's roofline object ("gemm_nt_nn",
"gemm_tn", "layernorm", "conv_module": launch-weighted means over the family's kernels), plus the commit and command the passes ran.
usage: python tools/pmc_summary.py <fetch.db> <write.db> <mfma.db> <steps> <out.json> [commit] [command]"""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def family(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"gemm_nt_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d)ELb(\d)", n)
    if m:
        kind = {"1": "conv2_fwd", "2": "conv2_dgrad"}.get(m.group(4), "gemm_nn" if m.group(5) == "1" else "gemm_nt")
        return f"{kind}[{m.group(2)}x{m.group(3)}]"
    m = re.search(r"big_nt_kernel<(\d+), (\d)>", n) or re.search(r"big_nt_kernelILi(\d+)ELi(\d)E", n)
    if m:
        return {"0": "gemm_nt_big", "1": "conv2_fwd_big", "2": "conv2_dgrad_big"}[m.group(2)] + f"[{32 * int(m.group(1))}x256]"
    m = re.search(r"(gemm_tn_grouped_kernel|cf_conv_bwd_kernel|cf_dwconv_kernel|gemm_tn_kernel|attn_bwd_fused_kernel|attn_bwd_kv_kernel|attn_bwd_q2_kernel|attn_bwd_q_kernel|attn_dropmask_kernel|attn_bwd_dpos3_kernel|attn_bwd_fin_kernel|attn_fwd4_kernel|"
                  r"attn_bwd_prep_kernel|attn_fwd_kernel|ln_bwd8_kernel|ln_fwd_kernel)", n)
    if m:
        return m.group(1)
    m = re.search(r"N12_GLOBAL__N_1\d+([a-z0-9_]+_kernel)", n)
    return m.group(1) if m else n.split("(")[0][:60]


def collect(db, counter):
    c = sqlite3.connect(db)
    out = defaultdict(lambda: [0, 0.0])
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        f = family(name)
        out[f][0] += 1
        out[f][1] += val
    return out


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
mfma, gui = collect(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES"), collect(sys.argv[3], "GRBM_GUI_ACTIVE")
steps = float(sys.argv[4])
res = {}
for f in sorted(set(fetch) | set(write) | set(mfma)):
    nf, kf = fetch.get(f, [0, 0.0])
    nw, kw = write.get(f, [0, 0.0])
    nm, busy = mfma.get(f, [0, 0.0])
    _, act = gui.get(f, [0, 0.0])
    n = max(nf, nw, nm)
    r = {"launches": n, "read_bytes": 2.0 * 1024.0 * kf / max(nf, 1), "write_bytes": 1024.0 * kw / max(nw, 1)}
    r["traffic_bytes"] = r["read_bytes"] + r["write_bytes"]
    if act > 0:
        r["mfma_util"] = busy / (4 * 256 * act / 8.0)
        r["mfma_busy_cycles"] = busy / max(nm, 1)
    res[f] = r
tot = sum(v["traffic_bytes"] * v["launches"] for v in res.values())
# families of bench.py's roofline object (emoasr_timer_read_ex): launch-weighted aggregates of their kernels
FAMILY_OF = {"gemm_nt_nn": ("gemm_nt[", "gemm_nn[", "gemm_nt_big["), "gemm_tn": ("gemm_tn_grouped_kernel", "gemm_tn_kernel"),
             "layernorm": ("ln_fwd_kernel", "ln_bwd8_kernel"),
             "conv_module": ("cf_conv_bwd_kernel", "cf_dwconv_kernel", "bn_bwd_sums_kernel", "bn_swish_fwd_kernel",
                             "bn_stats_finalize_kernel", "bn_bwd_fold_kernel", "dwconv_bwd_w_reduce_kernel")}
for fam, keys in FAMILY_OF.items():
    members = [v for k, v in res.items() if any(k.startswith(p) for p in keys)]
    n = sum(m["launches"] for m in members)
    if not n:
        continue
    agg = {"launches": n, "kernels": sorted(k for k in res if any(k.startswith(p) for p in keys))}
    for key in ("read_bytes", "write_bytes", "traffic_bytes", "mfma_util"):
        agg[key] = sum(m.get(key, 0.0) * m["launches"] for m in members) / n
    res[fam] = agg
json.dump({"note": "per launch, by kernel; reads = FETCH_SIZE x 2 (gfx950 correction), writes = WRITE_SIZE, mfma_util = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); three separate --pmc passes of "
                   "the bench command recorded under `command`",
           "commit": sys.argv[6] if len(sys.argv) > 6 else None, "command": sys.argv[7] if len(sys.argv) > 7 else None,
           "total_bytes_per_step": tot / steps, "kernels": res}, open(sys.argv[5], "w"), indent=1)
for f, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches"])[:24]:
    print(f"{f:34s} n={v['launches']:5d} read {v['read_bytes'] / 1e6:9.2f} MB  write {v['write_bytes'] / 1e6:9.2f} MB  "
          f"total/step {v['traffic_bytes'] * v['launches'] / steps / 1e6:9.1f} MB  mfma_util {v.get('mfma_util', 0):.3f}")
print(f"total {tot / steps / 1e9:.2f} GB per step")
