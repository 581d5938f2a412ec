"""HBM traffic per launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same
bench command, as MI355X_MICROARCH.md prescribes: both counters are in KB; on gfx950 FETCH_SIZE tallies
128-byte read requests at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte stores and f32 atomics.
usage: python tools/pmc_traffic.py <fetch.db> <write.db> <out.json>"""
import json, re, sqlite3, sys
from collections import defaultdict


def family(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"gemm_nt_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d)ELb(\d)", n)
    if m:
        kind = {"1": "conv2_fwd", "2": "conv2_dgrad"}.get(m.group(4), "gemm_nn" if m.group(5) == "1" else "gemm_nt")
        return f"{kind}[{m.group(2)}x{m.group(3)}]"
    m = re.search(r"(gemm_tn_grouped_kernel|gemm_tn_kernel|attn_bwd_dq2_kernel|attn_fwd_kernel|ln_bwd8_kernel|ln_fwd_kernel)", n)
    if m:
        return m.group(1)
    m = re.search(r"N12_GLOBAL__N_1\d+([a-z0-9_]+_kernel)", n)
    return m.group(1) if m else n.split("(")[0][:60]


def collect(db, counter):
    c = sqlite3.connect(db)
    out = defaultdict(lambda: [0, 0.0])
    for name, gz, val in c.execute("select kernel_name, grid_size_z, value from counters_collection where counter_name = ?", (counter,)):
        f = family(name)
        if f.startswith("gemm_nn") and gz > 1:
            f = f.replace("gemm_nn", "gemm_nn_batched")
        out[f][0] += 1
        out[f][1] += val * 1024.0
    return out


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
res = {}
for f in sorted(set(fetch) | set(write)):
    nf, bf = fetch.get(f, [0, 0.0])
    nw, bw = write.get(f, [0, 0.0])
    n = max(nf, nw)
    res[f] = {"launches": n, "read_bytes_per_launch": 2.0 * bf / max(nf, 1), "write_bytes_per_launch": bw / max(nw, 1)}
    res[f]["traffic_bytes_per_launch"] = res[f]["read_bytes_per_launch"] + res[f]["write_bytes_per_launch"]
tot = sum(v["traffic_bytes_per_launch"] * v["launches"] for v in res.values())
json.dump({"note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate --pmc passes of "
                   "`bench.py --steps 5 --warmup 2 --no-decode --no-cpu-baseline`; bytes per launch by kernel family",
           "total_bytes_per_step": tot / 7.0, "families": res}, open(sys.argv[3], "w"), indent=1)
for f, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:22]:
    print(f"{f:34s} n={v['launches']:5d} read {v['read_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB  total/step {v['traffic_bytes_per_launch'] * v['launches'] / 7e6:9.1f} MB")
print(f"total {tot / 7e9:.2f} GB per step")
