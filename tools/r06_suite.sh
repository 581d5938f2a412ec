cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_suite; mkdir -p $O
( time timeout 2400 python3 -m pytest tests -m gpu -q ) > $O/gputest.log 2>&1
grep -E "passed|failed|error" $O/gputest.log | tail -5; grep -E "^FAILED|^ERROR" $O/gputest.log | head -20
