"""Launch shapes from a rocprofv3 rocpd kernel trace: per kernel (and grid) the workgroups, threads, LDS and registers per
workgroup, how many workgroups a CU can hold at once (LDS / register / wave-slot limits of gfx950: 160 KB, 512 VGPRs per SIMD lane,
8 waves per SIMD... 32 per CU) and how many ROUNDS of resident workgroups the launch is -- launches of 1.0-1.5 or just over an integer
number of rounds are the ones with a tail.  usage: python tools/kshape.py <results.db> [min_us_per_call]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
def pick(*names):
    for n in names:
        if n in cols: return n
    return "0"
if "--cols" in sys.argv: print(cols)
# grid_* are in work-items; one line per kernel: launches of different grids are averaged (calls-weighted)
q = ("select name, avg(1.0 * grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z)), workgroup_x * workgroup_y * workgroup_z, "
     "max(lds_size, static_lds_size), vgpr_count, accum_vgpr_count, count(*), avg(end-start)/1e3 from kernels group by 1,3,4,5,6 order by 8*7 desc")
print(f"{'kernel':48s} {'calls':>6s} {'avg us':>8s} {'wgs':>7s} {'thr':>4s} {'lds KB':>7s} {'vgpr':>5s} {'wg/CU':>5s} {'rounds':>7s}")
for name, grid, wg, lds, vg, ag, n, us in c.execute(q):
    if us < min_us: continue
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"([a-z0-9_]+_kernel)", name)
    short = (m.group(1) if m else name[:40]) + ("" if "Li" not in name else " " + "x".join(re.findall(r"Li(\d+)E", name)[:3]))
    if not wg or not grid:
        continue
    wgs = int(grid)
    waves = (wg + 63) // 64
    regs = max(vg + ag, 1)
    per_simd = max(1, min(8, 512 // ((regs + 7) // 8 * 8)))
    by_reg = per_simd * 4 // waves if waves <= per_simd * 4 else 0
    by_lds = (160 * 1024) // lds if lds else 99
    by_wave = 32 // waves
    res = max(1, min(by_reg or 1, by_lds, by_wave))
    print(f"{short[:48]:48s} {n:6d} {us:8.1f} {wgs:7d} {wg:4d} {lds / 1024:7.1f} {regs:5d} {res:5d} {wgs / (256.0 * res):7.2f}")
