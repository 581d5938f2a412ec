cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_11; mkdir -p $O
( python -m pytest tests/test_ctc_beam_gpu.py -x -q ) > $O/tests.log 2>&1
tail -3 $O/tests.log
EMOASR_CTC_LM_CACHE=0 UTTS=2 timeout 600 python tools/ctc_beam_probe.py 2>/dev/null | tail -1 | tee $O/probe.txt
PROFILE=1 EMOASR_CTC_LM_CACHE=1 UTTS=2 timeout 600 python tools/ctc_beam_probe.py 2>/dev/null | tee -a $O/probe.txt | head -30
