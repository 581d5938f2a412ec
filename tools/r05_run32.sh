cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_32; mkdir -p $O
( python -m pytest tests/test_stacked_gpu.py tests/test_stacked_oracle_gpu.py tests/test_model_gpu.py -x -q ) > $O/tests.log 2>&1
tail -n 3 $O/tests.log
for i in 1 2; do
EMOASR_OPTIONS="stack_launch=0" python3 tools/f32_leg.py --split --stacked 2>&1 | grep -v amdgpu.ids | sed 's/^/per-seg launches: /'
python3 tools/f32_leg.py --split --stacked 2>&1 | grep -v amdgpu.ids | sed 's/^/one launch:       /'
done | tee $O/ab.txt
