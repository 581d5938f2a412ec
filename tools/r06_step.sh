# bench step (no decode legs) + kernel table;  usage: bash tools/r06_step.sh <tag>
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_$1; mkdir -p $O
CMD="python3 bench.py --steps 6 --warmup 2 --no-decode --no-cpu-baseline"
$CMD > $O/bench_nodecode.json 2> $O/bench.err
python3 -c "import json;d=json.load(open('$O/bench_nodecode.json'));print('VALUE',d['value'],d['ms_per_step']);print({k:round(v['ms'],3) for k,v in d['families'].items() if isinstance(v,dict)})"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
python3 tools/kstats.py $O/kt/kt_results.db 8 $O/kernel_stats.csv > $O/kstats.txt 2>&1
python3 tools/kshape.py $O/kt/kt_results.db 18 > $O/launch_shapes.txt 2>&1
rm -rf $O/kt
head -36 $O/kstats.txt | cut -c1-150
