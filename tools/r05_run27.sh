cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_27; mkdir -p $O
EMOASR_BENCH_TRACE=1 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -E "parity|headline" $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], {k: v for k, v in d.items() if k.startswith('parity_mode') or k.startswith('f32_')})"
EMOASR_BENCH_TRACE=1 PYTORCH_HIP_ALLOC_CONF=expandable_segments:True python3 bench.py --no-cpu-baseline --no-decode > $O/bench2.json 2> $O/bench2.err
grep -E "headline" $O/bench2.err; python3 -c "
import json; d=json.load(open('$O/bench2.json')); print('expandable', d['value'], d['ms_per_step'])"
EMOASR_BENCH_TRACE=1 python3 bench.py --no-cpu-baseline --no-decode > $O/bench3.json 2> $O/bench3.err
grep -E "headline" $O/bench3.err; python3 -c "
import json; d=json.load(open('$O/bench3.json')); print('default', d['value'], d['ms_per_step'])"
