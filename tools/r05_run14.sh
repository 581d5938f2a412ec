cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_15; mkdir -p $O
( EMOASR_FORCE_SPLIT=1 python3 -m pytest tests -m gpu -q -k "not bf16" ) > $O/forced.log 2>&1
tail -40 $O/forced.log | grep -v "^$" | tail -40
