cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_6; mkdir -p $O
( time python -m pytest tests/test_streams_gpu.py -x -q ) > $O/streams.log 2>&1
tail -5 $O/streams.log
( time python -m pytest tests -m gpu -x -q ) > $O/gputest.log 2>&1
tail -5 $O/gputest.log
