cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_5; mkdir -p $O
( python -m pytest tests/test_model_gpu.py tests/test_stacked_gpu.py -x -q ) > $O/tests.log 2>&1
tail -2 $O/tests.log
run() { EMOASR_DGRAD_NT2=$1 EMOASR_OPTIONS="$2" python bench.py --no-decode --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['families']
print('nt2=$1 $2'.ljust(34), round(d['ms_per_step'],2), 'ms/step  ', '  '.join(k.replace('_kernel','')+' '+str(round(v['ms'],2)) for k,v in f.items() if isinstance(v,dict)))"; }
for rep in 1 2 3; do
run 0 ""
run 1 ""
run 1 "attn_side_prio=1"
run 0 "attn_side_prio=1"
done | tee $O/ab.txt
