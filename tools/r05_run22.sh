cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_22; mkdir -p $O
( EMOASR_CPP_WGRAD_SIDE=1 python -m pytest tests/test_model_gpu.py tests/test_stacked_gpu.py tests/test_stacked_oracle_gpu.py tests/test_train_gpu.py tests/test_fullsize_gpu.py tests/test_00_dp_two_process_gpu.py tests/test_00_bench_two_ranks_gpu.py tests/test_l4_gpu.py -x -q ) > $O/tests.log 2>&1
tail -n 3 $O/tests.log
run() { EMOASR_CPP_WGRAD_SIDE=$1 python bench.py --no-decode --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['families']
print('wgrad_side=$1'.ljust(16), round(d['ms_per_step'],2), 'ms/step  ', round(d['value']), '  '.join(k.replace('_kernel','')+' '+str(round(v['ms'],2)) for k,v in f.items() if isinstance(v,dict)))"; }
for rep in 1 2 3; do
run 0
run 1
done | tee $O/ab.txt
