set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4k
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib/exp
for i in 1 2; do for f in $L/libw43_0.so $L/libw43_6.so; do
  echo "== $(basename $f)"
  VDIFF_HIP_LIB=$f python tests/perf_wino43.py 2>&1 | grep "FORWARD" | head -3 | sed 's/err [0-9.e+-]*//g'
done; done > $OUT/exp6.txt 2>&1
grep -v "^+" $OUT/exp6.txt | cut -c1-220
