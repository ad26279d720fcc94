R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_conv_train.py -x -q 2>&1 | tail -3
for v in "" ct_pp0; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export SPKDIFF_LIB=$R/spiking-diffusion_amd/spkdiff/variants/$v.so; fi
  python tools/conv_train_time.py 512 nolib 2>/dev/null
done
