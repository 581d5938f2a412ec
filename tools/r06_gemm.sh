cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_gemm; mkdir -p $O
BLAS=1 EPI=1 M=35145 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_gemm_35k_rows.txt
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  M=35145 rocprofv3 --kernel-trace --pmc $grp -d $O/c$i -o c$i -- python3 tools/gemm_bench.py > /dev/null 2>&1
  python3 tools/pmc_counters.py $O/c$i/c${i}_results.db gemm_nt_kernel 2>&1 | head -40
  rm -rf $O/c$i
done > $O/gemm_counters.txt 2>&1
head -120 $O/gemm_counters.txt
