R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_conv_train.py -x -q 2>&1 | tail -2
for v in "" ct_fb0 "" ct_fb0; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export SPKDIFF_LIB=$R/spiking-diffusion_amd/spkdiff/variants/$v.so; else unset SPKDIFF_LIB; fi
  python tools/conv_train_time.py 512 nolib 2>/dev/null | grep -E "conv2|convT1|convT2|total"
done
