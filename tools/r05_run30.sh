cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_30; mkdir -p $O
UTTS=2 rocprofv3 --kernel-trace --stats -d $O/kx -o kx -- python3 tools/ctc_beam_probe.py > $O/kx.log 2>&1
python3 tools/kstats.py $O/kx/kx_results.db 1 $O/ctcbeam_kstats.csv 2>&1 | head -n 30 | cut -c1-150
rm -rf $O/kx
