cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_4; mkdir -p $O
( time python -m pytest tests/test_model_gpu.py tests/test_stacked_oracle_gpu.py tests/test_l4_gpu.py tests/test_kd_gpu.py -x -q ) > $O/tests.log 2>&1
tail -3 $O/tests.log
CMD="python3 bench.py --steps 6 --warmup 2 --no-decode --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
python3 tools/kstats.py $O/kt/kt_results.db 8 $O/kernel_stats.csv > $O/kstats.txt 2>&1
python3 tools/kseq.py $O/kt/kt_results.db $O/kseq.txt > /dev/null 2>&1
python3 tools/kshape.py $O/kt/kt_results.db 18 > $O/launch_shapes.txt 2>&1
rm -rf $O/kt
