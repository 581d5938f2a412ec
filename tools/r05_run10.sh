cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_10; mkdir -p $O
( python -m pytest tests/test_l4_gpu.py tests/test_abi.py -x -q ) > $O/tests.log 2>&1
tail -4 $O/tests.log
for o in "" "rnnt_beam_mfma=0" ""; do echo "opt=$o"; EMOASR_OPTIONS="$o" python tools/l4_beam_prof.py 2>/dev/null | head -1; done | tee $O/beam.txt
D=$O/kt; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o k -- python3 tools/l4_beam_prof.py > $O/kt.log 2>&1
python3 tools/kstats.py $D/k_results.db 1 > $O/kstats.txt 2>&1
rm -rf $D
grep -i "beam\|gemm_nt\|copy" $O/kstats.txt | head
