R=$GRAFT_REPO_ROOT
cd $R
for v in "" ct_dbg32 ct_dbg64 ct_dbg128 ct_dbg224; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export SPKDIFF_LIB=$R/spiking-diffusion_amd/spkdiff/variants/$v.so; fi
  python tools/conv_train_time.py 512 nolib 2>/dev/null | grep -E "conv2|convT2"
done
