# round-6 baseline on the cleaned tree: GPU suite, bench step (no decode legs), kernel table
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_base; mkdir -p $O
( time timeout 1500 python3 -m pytest tests -m gpu -q -x ) > $O/gputest.log 2>&1
tail -5 $O/gputest.log
CMD="python3 bench.py --steps 6 --warmup 2 --no-decode --no-cpu-baseline"
$CMD > $O/bench_nodecode.json 2> $O/bench.err
cat $O/bench_nodecode.json | cut -c1-600
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
python3 tools/kstats.py $O/kt/kt_results.db 8 $O/kernel_stats.csv > $O/kstats.txt 2>&1
python3 tools/kshape.py $O/kt/kt_results.db 18 > $O/launch_shapes.txt 2>&1
rm -rf $O/kt
head -30 $O/kernel_stats.csv | cut -c1-160
