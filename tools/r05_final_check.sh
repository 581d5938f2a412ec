cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
( time python3 -m pytest tests -m gpu -x -q ) > $O/gputest.log 2>&1
tail -3 $O/gputest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python3 bench.py > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v for k, v in d.items() if k.startswith('parity_mode') or k.startswith('f32_')}, d['l4_rnnt']['beam4_rtf'], d['ctc_beam'])"
