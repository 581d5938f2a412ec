set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${EV_TAG:-r04_c}; O=gpurun_out/${TAG}_ev; mkdir -p $O
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
python3 bench.py > $O/${TAG}_bench.json 2> $O/bench.err
