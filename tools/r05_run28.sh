cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_28; mkdir -p $O
( python -m pytest tests/test_train_gpu.py -x -q -s -k converges ) > $O/tests.log 2>&1
grep -E "measured|passed|failed|Error|assert" $O/tests.log | head -n 20
