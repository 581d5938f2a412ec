# Round-6 evidence set (run on the GPU box through gpurun): kernel table / sequence / launch shapes of the bench step, the three
# counter passes behind roofline.traffic / mfma_util, the attention kernels' SQ instruction counters, the f32x3 step's kernel
# table, the full GPU suite and the bench line.   EV_COMMIT=<sha> bash tools/r06_evidence.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=r06; O=gpurun_out/${TAG}_ev; mkdir -p $O
CMD="python3 bench.py --steps 6 --warmup 2 --no-decode --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
python3 tools/kstats.py $O/kt/kt_results.db 8 $O/${TAG}_kernel_stats.csv > $O/kstats.txt 2>&1
python3 tools/kseq.py $O/kt/kt_results.db $O/${TAG}_kseq.txt > /dev/null 2>&1
python3 tools/kshape.py $O/kt/kt_results.db 18 > $O/${TAG}_launch_shapes.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/p1 -o p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/p2 -o p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p3 -o p3 -- $CMD > $O/p3.log 2>&1
python3 tools/pmc_summary.py $O/p1/p1_results.db $O/p2/p2_results.db $O/p3/p3_results.db 8 $O/${TAG}_pmc.json "${EV_COMMIT:-unknown}" "$CMD (three separate rocprofv3 --pmc passes)" > $O/pmc.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3 $O/kt
# attention kernels: instruction counters per launch (judge item 2a)
export B=110 T=320 MODES=fused   # (five stacked micro-batches' worth of one layer; tools/attn_bench.py)
echo "# rocprofv3 --pmc <group> -- python3 tools/attn_bench.py (B 110, T' 320, bf16, H 4, dropout 0.1): per-launch averages" > $O/${TAG}_attn_counters.txt
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/a$i -o a$i -- python3 tools/attn_bench.py > $O/a$i.log 2>&1
  python3 tools/pmc_counters.py $O/a$i/a${i}_results.db attn_ >> $O/${TAG}_attn_counters.txt 2>&1
  rm -rf $O/a$i
done
unset B T MODES
# the f32x3 (parity) step's kernel tables: one micro-batch per step, and five stacked
rocprofv3 --kernel-trace --stats -d $O/kx -o kx -- python3 tools/f32_leg.py --split > $O/kx.log 2>&1
python3 tools/kstats.py $O/kx/kx_results.db 9 $O/${TAG}_f32x3_kernel_stats.csv > $O/f32x3_kstats.txt 2>&1
rm -rf $O/kx
rocprofv3 --kernel-trace --stats -d $O/ky -o ky -- python3 tools/f32_leg.py --split --stacked > $O/ky.log 2>&1
python3 tools/kstats.py $O/ky/ky_results.db 9 $O/${TAG}_f32x3_stacked_kernel_stats.csv > $O/f32x3_stacked_kstats.txt 2>&1
rm -rf $O/ky
# in-kernel phase stamps of the attention kernels (a -DEMO_ATTN_STAMP build variant: wave 0 of workgroup 0, cycles per key / query step)
if [ -f emoasr_amd/build/libemoasr_hip_stamp.so ]; then
  ( export EMOASR_HIP_LIB=$GRAFT_REPO_ROOT/emoasr_amd/build/libemoasr_hip_stamp.so
    for bt in "110 320" "36 208" "11 651"; do set -- $bt; B=$1 T=$2 MODES=fused python3 tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done ) > $O/${TAG}_attn_phases.txt 2>&1
fi
MODES=fused python3 tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_attn_bench.txt
BLAS=1 EPI=1 M=35145 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_35k_rows.txt
( time python3 -m pytest tests -m gpu -q ) > $O/${TAG}_gputest.log 2>&1
python3 bench.py > $O/${TAG}_bench.json 2> $O/bench.err
tail -3 $O/${TAG}_gputest.log
