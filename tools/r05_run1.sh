cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_0
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r05_0/gputest.log 2>&1
python bench.py > gpurun_out/r05_0/bench.json 2> gpurun_out/r05_0/bench.err
D=gpurun_out/r05_0/f32kt; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o k -- python3 tools/f32_leg.py > gpurun_out/r05_0/f32_leg.log 2>&1
python3 tools/kstats.py $D/k_results.db 9 gpurun_out/r05_0/f32_kernel_stats.csv > gpurun_out/r05_0/f32_kstats.txt 2>&1
rm -rf $D
tail -3 gpurun_out/r05_0/gputest.log
