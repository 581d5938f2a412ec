# same-box A/B of the bench step: alternates option sets;  usage: bash tools/ab_step.sh <tag> "<optsA>" "<optsB>" [rounds]
# (option sets: EMOASR_OPTIONS strings, "-" = none; a leading "lib:<variant>" selects emoasr_amd/build/libemoasr_hip_<variant>.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_$1; mkdir -p $O
N=${4:-3}
run() {  # $1 = option string
  local opts="$1" libv=""
  if [[ "$opts" == lib:* ]]; then libv="${opts%%,*}"; libv="${libv#lib:}"; opts="${opts#lib:$libv}"; opts="${opts#,}"; fi
  [ "$opts" = "-" ] && opts=""
  # items in capitals are environment variables of the Python side (NAME=value), the rest library options
  local envs="" lo=""
  IFS=',' read -ra items <<< "$opts"
  for it in "${items[@]}"; do
    if [[ "$it" =~ ^[A-Z_0-9]+= ]]; then envs="$envs $it"; else lo="${lo:+$lo,}$it"; fi
  done
  opts="$lo"
  ( [ -n "$envs" ] && export $envs
    [ -n "$libv" ] && export EMOASR_HIP_LIB=$GRAFT_REPO_ROOT/emoasr_amd/build/libemoasr_hip_$libv.so
    EMOASR_OPTIONS="$opts" python3 bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys;d=json.loads(sys.stdin.read());f=d['families'];print('%.3f ms/step  %.3f M/s  '%(d['ms_per_step'],d['value']/1e6)+' '.join('%s %.2f'%(k.replace('_kernel',''),v['ms']) for k,v in f.items() if isinstance(v,dict)))" )
}
for i in $(seq $N); do
  echo "A [$2]: $(run "$2")"
  echo "B [$3]: $(run "$3")"
done | tee $O/ab.txt
