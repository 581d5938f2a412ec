# A/B of launch-policy options inside the training step: ms per optimizer step and the family timers of bench.py
# usage: bash tools/opt_ab.sh "opt=v,opt=v" "..." ...   (an empty string = defaults)
for o in "$@"; do
  EMOASR_OPTIONS="$o" python bench.py --no-decode --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['families']
print('$o'.ljust(34), round(d['ms_per_step'],2), 'ms/step  ', '  '.join(k.replace('_kernel','')+' '+str(round(v['ms'],2)) for k,v in f.items() if isinstance(v,dict)))"
done
