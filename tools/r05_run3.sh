cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_2; mkdir -p $O
( time python -m pytest tests/test_split_gpu.py -x -q ) > $O/split_test.log 2>&1
( time python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "against_oracle" ) > $O/fullsize.log 2>&1
python tools/f32_leg.py --split > $O/f32x3_leg.txt 2>&1
D=$O/kt; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o k -- python3 tools/f32_leg.py --split > $O/f32x3_leg_prof.log 2>&1
python3 tools/kstats.py $D/k_results.db 9 $O/f32x3_kernel_stats.csv > $O/f32x3_kstats.txt 2>&1
rm -rf $D
tail -5 $O/split_test.log; grep "measured\|passed\|failed" $O/fullsize.log | tail -40; tail -2 $O/f32x3_leg.txt
