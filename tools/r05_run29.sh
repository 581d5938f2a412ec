cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_29; mkdir -p $O
( python -m pytest tests/test_kd_gpu.py tests/test_l4_gpu.py -x -q ) > $O/tests.log 2>&1
tail -n 4 $O/tests.log
