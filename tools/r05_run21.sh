cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_21; mkdir -p $O
PROFILE=1 UTTS=3 python3 tools/ctc_beam_probe.py 2>&1 | grep -v amdgpu.ids > $O/probe.txt
head -n 40 $O/probe.txt | cut -c1-170
