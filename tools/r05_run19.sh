cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_19; mkdir -p $O
for d in bf16 f32 f32x3; do echo "== $d"; DT=$d python3 tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done > $O/attn.log
cat $O/attn.log
DT=f32x3 B=40 T=170 rocprofv3 --kernel-trace --stats -d $O/kx -o kx -- python3 tools/attn_bench.py > $O/kx.log 2>&1
python3 tools/kstats.py $O/kx/kx_results.db 1 $O/attn_f32x3_kstats.csv 2>&1 | head -n 14 | cut -c1-150
rm -rf $O/kx
