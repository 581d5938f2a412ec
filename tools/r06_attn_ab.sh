# attention entry points under several library variants on one box: usage bash tools/r06_attn_ab.sh "<variants>" (default = the tree's library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_attn_ab; mkdir -p $O
for v in $1; do
  for bt in "110 320" "36 208"; do
    set -- $bt
    if [ "$v" = default ]; then unset EMOASR_HIP_LIB; else export EMOASR_HIP_LIB=$GRAFT_REPO_ROOT/emoasr_amd/build/libemoasr_hip_$v.so; fi
    echo "== $v"; B=$1 T=$2 MODES=fused python3 tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
    if [ "$v" != stamp ]; then
      B=$1 T=$2 MODES=fused rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 tools/attn_bench.py > /dev/null 2>&1
      python3 tools/kstats.py $O/kt/kt_results.db 1 2>&1 | grep -i "attn" | cut -c1-120; rm -rf $O/kt
    fi
  done
done
