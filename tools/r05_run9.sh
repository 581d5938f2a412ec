cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_9; mkdir -p $O
python tools/l4_beam_prof.py > $O/prof.txt 2>&1
D=$O/kt; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o k -- python3 tools/l4_beam_prof.py > $O/kt.log 2>&1
python3 tools/kstats.py $D/k_results.db 1 > $O/kstats.txt 2>&1
rm -rf $D
head -40 $O/prof.txt
grep -i "beam\|gemm_nt\|copy\|Memcpy" $O/kstats.txt | head -20
