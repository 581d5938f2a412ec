# conv1_wgrad row-chunk A/B: the default library (32 t1 rows per workgroup) against two variants (16, 8), per-kernel time from a short trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_24; mkdir -p $O
for v in base t16 t8 base t16; do
  if [ $v = base ]; then unset EMOASR_HIP_LIB; else export EMOASR_HIP_LIB=$GRAFT_REPO_ROOT/emoasr_amd/build/libemoasr_hip_$v.so; fi
  rocprofv3 --kernel-trace --stats -d $O/kx_$v -o kx -- python3 bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline > $O/kx_$v.log 2>&1
  echo "== $v: $(grep -o '"ms_per_step": [0-9.]*' $O/kx_$v.log | head -n 1)"
  python3 tools/kstats.py $O/kx_$v/kx_results.db 6 $O/ks_$v.csv 2>/dev/null | grep -E "conv1_wgrad|total kernel" | cut -c1-120
  rm -rf $O/kx_$v
done 2>&1 | tee $O/ab.txt
