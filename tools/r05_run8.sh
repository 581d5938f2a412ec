cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_8; mkdir -p $O
( python -m pytest tests/test_l4_gpu.py tests/test_abi.py -x -q ) > $O/tests.log 2>&1
tail -4 $O/tests.log
for f in 1 0 1 0; do echo "fused=$f"; EMOASR_RNNT_BEAM_FUSED=$f python tools/l4_beam_prof.py 2>/dev/null | head -1; done | tee $O/beam.txt
