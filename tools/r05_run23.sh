cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_23; mkdir -p $O
( python -m pytest tests/test_model_gpu.py tests/test_stacked_gpu.py -x -q ) > $O/tests.log 2>&1
tail -n 12 $O/tests.log
