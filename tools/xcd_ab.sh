R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fp6 or mfma" 2>&1 | tail -2
timeout 300 python tools/fp6_variants.py spiking-diffusion_amd/spkdiff/variants/libspkdiff_lin.so spiking-diffusion_amd/spkdiff/variants/libspkdiff_xcd.so spiking-diffusion_amd/spkdiff/variants/libspkdiff_lin.so spiking-diffusion_amd/spkdiff/variants/libspkdiff_xcd.so 2>&1
cd /tmp && export TMPDIR=/tmp
export SPKDIFF_LIB=$R/spiking-diffusion_amd/spkdiff/variants/libspkdiff_xcd.so
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof6/fetch_xcd -- python $R/tools/fp6_one.py 512 256 13 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof6/write_xcd -- python $R/tools/fp6_one.py 512 256 13 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
for d in ("fetch_xcd","write_xcd"):
    f=glob.glob(f"{R}/gpurun_out/prof6/{d}/runc/*_counter_collection.csv")[0]
    v=sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "conv3x3_fp6" in r["Kernel_Name"])
    print(d, v[len(v)//2], len(v))
PY
