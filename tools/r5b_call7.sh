R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_conv_train.py -x -q 2>&1 | tail -2
python tools/conv_train_time.py 512 nolib 2>/dev/null
python tools/conv_train_time.py 512 nolib 2>/dev/null | tail -1
