cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_16; mkdir -p $O
( python -m pytest tests/test_split_gpu.py -x -q ) > $O/tests.log 2>&1
tail -2 $O/tests.log
for rep in 1 2; do
for o in "split_tile=3" "" "split_min128=256" "split_min128=128" "split_min128=1"; do
  echo -n "opt=$o  " ; EMOASR_OPTIONS="$o" python tools/f32_leg.py --split 2>/dev/null | tail -1
done
done | tee $O/ab.txt
