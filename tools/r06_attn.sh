# attention iteration loop: kernel tests, then per-kernel times of the attention entry points under rocprofv3
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_attn; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_split_gpu.py -m gpu -q -x -k "attention or attn" > $O/test.log 2>&1
tail -6 $O/test.log
MODES=fused python3 tools/attn_bench.py > $O/bench.txt 2>&1; cat $O/bench.txt
export B=110 T=320 MODES=fused
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 tools/attn_bench.py > $O/kt.log 2>&1
python3 tools/kstats.py $O/kt/kt_results.db 1 $O/kernel_stats.csv 2>&1 | grep -i "attn" 
rm -rf $O/kt
