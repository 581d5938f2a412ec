# flakiness check: the whole GPU suite three times in a row (fresh processes), smoke in between
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_31; mkdir -p $O
for i in 1 2 3; do
  python3 -m pytest tests -m gpu -q -x > $O/gputest_$i.log 2>&1
  grep -E "passed|failed|error" $O/gputest_$i.log | tail -n 2
  python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
done
