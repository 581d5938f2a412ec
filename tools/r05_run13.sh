cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_14; mkdir -p $O
( python -m pytest tests/test_ctc_beam_gpu.py tests/test_l3_gpu.py tests/test_stacked_gpu.py tests/test_stacked_oracle_gpu.py tests/test_fullsize_l3_l4_gpu.py -x -q ) > $O/tests.log 2>&1
tail -4 $O/tests.log
UTTS=3 python tools/ctc_beam_probe.py 2>/dev/null | tail -1 | tee $O/probe.txt
EMOASR_LM_GRAPH=0 UTTS=3 python tools/ctc_beam_probe.py 2>/dev/null | tail -1 | tee -a $O/probe.txt
python bench.py --no-decode --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/probe.txt
