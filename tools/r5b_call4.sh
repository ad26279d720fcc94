R=$GRAFT_REPO_ROOT
cd $R
for v in "" ct_big2048 ct_big512; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export SPKDIFF_LIB=$R/spiking-diffusion_amd/spkdiff/variants/$v.so; fi
  python tools/conv_train_time.py 512 nolib 2>/dev/null
done
