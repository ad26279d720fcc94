R=$GRAFT_REPO_ROOT
cd $R
bash tools/train_other_prof.sh 2>&1 | grep -v "^W2026"
python tools/conv_train_time.py 512 2>/dev/null
