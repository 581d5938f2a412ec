# A/B of the grouped weight-gradient launch policies inside the training step (family timer of bench.py)
for o in "" "tn_place=1" "tn_group_kb=2,tn_group_blocks=512" "tn_group_blocks=640" "tn_group_blocks=896" ""; do
  EMOASR_OPTIONS="$o" python bench.py --no-decode --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['families']['gemm_tn']
print('$o'.ljust(40), round(d['ms_per_step'],2), 'ms/step   gemm_tn family', round(f['ms'],2), 'ms', f['calls'], 'calls')"
done
