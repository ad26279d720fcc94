# round 4: last-position split threshold -- dense, elimination and lists modes, per variant, two passes
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for pass in 1 2; do
  for lib in "$@"; do
    echo "== pass $pass $lib"
    for mode in dense elim elim+lists4; do SPKDIFF_LIB=$R/$lib python $R/tools/listed_time.py 256 3 $mode 2>/dev/null; done
  done
done
