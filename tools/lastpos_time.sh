# times conv3x3_fp6_lastpos_kernel at the four denoiser shapes under rocprofv3, for every SPKDIFF_LIB given
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
export SPKDIFF_LIB=$R/$lib
for shp in "512 256" "256 512" "256 128" "128 64"; do
d=$R/gpurun_out/lp/$(basename $lib .so)_$(echo $shp | tr ' ' _); rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/tools/fp6_one.py $shp 10 > /dev/null 2>&1
python - <<PY
import csv, glob
f=glob.glob("$d/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "lastpos" in r["Name"]: print("$(basename $lib .so)", "$shp", r["Calls"], r["AverageNs"])
PY
done
done
