cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_7; mkdir -p $O
for i in 1 2 3 4 5; do python -m pytest tests/test_streams_gpu.py -x -q 2>&1 | tail -1; done > $O/streams.log 2>&1
cat $O/streams.log
( time python -m pytest tests/test_ops_gpu.py tests/test_streams_gpu.py tests/test_l4_gpu.py tests/test_fullsize_l3_l4_gpu.py -x -q -s ) > $O/t2.log 2>&1
grep "measured L4\|passed\|failed" $O/t2.log | tail -50
