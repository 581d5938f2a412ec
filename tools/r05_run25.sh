cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_25; mkdir -p $O
( python -m pytest tests/test_ctc_beam_gpu.py -x -q ) > $O/tests.log 2>&1
tail -n 5 $O/tests.log
for v in 1 0 1 0; do EMOASR_CTC_BEAM_NATIVE=$v UTTS=3 python3 tools/ctc_beam_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/native=$v  /"; done | tee $O/probe.txt
EMOASR_CTC_BEAM_NATIVE=1 PROFILE=1 UTTS=3 python3 tools/ctc_beam_probe.py 2>&1 | grep -v amdgpu.ids | head -n 28 | cut -c1-160 >> $O/probe.txt
tail -n 26 $O/probe.txt
