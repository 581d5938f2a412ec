# per-kernel device times of a python command under rocprofv3 --kernel-trace (run on the GPU box):
#   bash tools/kprof.sh <out file under gpurun_out> <steps to divide by> <grep filter> <python script + args ...>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; STEPS=$2; F=$3; shift 3
D=$GRAFT_REPO_ROOT/gpurun_out/kprof_tmp; rm -rf $D; mkdir -p $D; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $D -o k -- python3 "$@" > $D/log.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $D/k_results.db $STEPS 2>/dev/null | grep -i "$F" >> $OUT
rm -rf $D
