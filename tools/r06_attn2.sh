cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_attn; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_split_gpu.py tests/test_l3_gpu.py -m gpu -q -x -k "attention or attn or decode" > $O/test.log 2>&1
tail -4 $O/test.log
bash tools/r06_attn_ab.sh "default stamp"
