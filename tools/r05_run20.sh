cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_20; mkdir -p $O
L4_STEPS=4 rocprofv3 --kernel-trace --stats -d $O/kx -o kx -- python3 tools/l4_leg.py > $O/kx.log 2>&1
tail -n 2 $O/kx.log
python3 tools/kstats.py $O/kx/kx_results.db 6 $O/l4_kstats.csv 2>&1 | head -n 45 | cut -c1-160
rm -rf $O/kx
