# round 5, second session: the native training convolutions -- tests, then the VQ-VAE training iteration's kernel table
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_conv_train.py -x -q 2>&1 | tail -25
timeout 600 bash tools/train_other_prof.sh 2>&1 | tail -60
