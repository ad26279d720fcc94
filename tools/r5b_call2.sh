# conv_train timing: the default build and the ablation variants on the largest layer
R=$GRAFT_REPO_ROOT
cd $R
python tools/conv_train_time.py 512
for d in 1 2 4 8 16; do
  echo "== CT_DBG=$d"
  CT_ONLY=dec.convT2 SPKDIFF_LIB=$R/spiking-diffusion_amd/spkdiff/variants/ct_dbg$d.so python tools/conv_train_time.py 512 nolib | head -1
done
