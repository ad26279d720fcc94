R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do timeout 300 python tools/fp6_variants.py spiking-diffusion_amd/spkdiff/variants/*.so; done
cd /tmp && export TMPDIR=/tmp
for shp in "512 256" "256 512" "256 128" "128 64"; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lp/$(echo $shp | tr ' ' _) -- python $R/tools/fp6_one.py $shp 10 > /dev/null 2>&1
python - <<PY
import csv, glob
f=glob.glob("$R/gpurun_out/lp/$(echo $shp | tr ' ' _)/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "fp6" in r["Name"] and "pack" not in r["Name"] and "spikes" not in r["Name"]: print("$shp", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
done
