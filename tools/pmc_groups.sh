# SQ counter groups (one rocprofv3 --pmc pass each) for the kernels whose name contains FILTER; run on the GPU box:
#   bash tools/pmc_groups.sh <out-dir under gpurun_out> <kernel-name filter> <python script + args ...>
# Only the per-kernel averages travel back (the databases are tens of MB each).
O=$GRAFT_REPO_ROOT/gpurun_out/$1; F=$2; shift 2
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o p$i -- python3 "$@" > $O/p$i.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_counters.py $O/p$i/p${i}_results.db "$F" >> $O/counters.txt 2>&1
  [ $i = 1 ] && python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/p1/p1_results.db 1 2>/dev/null | grep "$F" >> $O/counters.txt
  rm -rf $O/p$i
done
