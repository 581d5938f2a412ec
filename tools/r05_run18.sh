cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_18; mkdir -p $O
python3 tools/f32_leg.py --split > $O/leg.log 2>&1
python3 tools/f32_leg.py --split --stacked >> $O/leg.log 2>&1
python3 tools/f32_leg.py --stacked >> $O/leg.log 2>&1
cat $O/leg.log | tail -n 8
rocprofv3 --kernel-trace --stats -d $O/kx -o kx -- python3 tools/f32_leg.py --split --stacked > $O/kx.log 2>&1
python3 tools/kstats.py $O/kx/kx_results.db 9 $O/f32x3_stacked_kernel_stats.csv > $O/f32x3_stacked_kstats.txt 2>&1
rm -rf $O/kx
head -n 40 $O/f32x3_stacked_kstats.txt | cut -c1-170
