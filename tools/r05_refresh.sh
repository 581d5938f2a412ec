# refresh of the round-5 evidence after the last kernel change (split-product tile rule): the f32x3 step's kernel table and the bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_ev2; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kx -o kx -- python3 tools/f32_leg.py --split > $O/kx.log 2>&1
python3 tools/kstats.py $O/kx/kx_results.db 9 $O/r05_f32x3_kernel_stats.csv > $O/f32x3_kstats.txt 2>&1
rm -rf $O/kx
python3 bench.py > $O/r05_bench.json 2> $O/bench.err
python3 -c "
import json; d=json.load(open('$O/r05_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity_mode_frames_per_s'], d['f32x3_vs_f32'], d['l4_rnnt']['beam4_rtf'], d['ctc_beam'])"
head -12 $O/f32x3_kstats.txt | cut -c1-150
