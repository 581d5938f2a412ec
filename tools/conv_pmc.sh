# SQ counters of the fused convolution-module kernels at the stacked shape (one counter group per pass; run on the GPU box)
O=gpurun_out/conv_pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export B=110
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $R/$O/p$i -o p$i -- python3 $R/tools/convmod_bench.py > $R/$O/p$i.log 2>&1
  python3 $R/tools/pmc_counters.py $R/$O/p$i/p${i}_results.db cf_ >> $R/$O/counters.txt 2>&1
  [ $i = 1 ] && python3 $R/tools/kstats.py $R/$O/p1/p1_results.db 1 2>/dev/null | grep "cf_" >> $R/$O/counters.txt
  rm -rf $R/$O/p$i    # (the databases are tens of MB each; only the summary travels back)
done
