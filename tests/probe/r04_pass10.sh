set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4j
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib/exp
for f in $L/libw43_0.so $L/libw43_1.so $L/libw43_2.so $L/libw43_3.so $L/libw43_4.so $L/libw43_5.so; do
  echo "== $(basename $f)"
  VDIFF_HIP_LIB=$f python tests/perf_wino43.py 2>&1 | grep "FORWARD" | head -2 | sed 's/err [0-9.e+-]*//g'
done > $OUT/exp.txt 2>&1
grep -v "^+" $OUT/exp.txt | cut -c1-220
