cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_17; mkdir -p $O
timeout 900 python3 -m pytest tests/test_stacked_gpu.py tests/test_stacked_oracle_gpu.py -x -q -m gpu -s 2>&1 | tail -40 > $O/stacked.log
timeout 1200 python3 -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_split_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/f32.log
tail -n 5 $O/stacked.log; tail -n 5 $O/f32.log
