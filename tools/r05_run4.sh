cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_3; mkdir -p $O
( time python -m pytest tests/test_split_gpu.py -x -q ) > $O/split_test.log 2>&1
tail -3 $O/split_test.log
for rep in 1 2; do
for o in "" "split_tile=1" "split_kb=2" "split_tile=1,split_kb=2" "split_tile=3"; do
  echo "== $o" >> $O/ab.txt
  EMOASR_OPTIONS="$o" python tools/f32_leg.py --split 2>/dev/null | tail -1 >> $O/ab.txt
done
done
cat $O/ab.txt
