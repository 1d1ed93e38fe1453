set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4r
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib/exp
for i in 1 2; do
  echo "== default"; python tests/perf_wino43.py 2>&1 | grep "FORWARD" | head -3 | sed 's/err [0-9.e+-]*//g'
  for a in 2 1 3; do echo "== aux $a"; VDIFF_HIP_LIB=$L/libvd_aux$a.so python tests/perf_wino43.py 2>&1 | grep "FORWARD" | head -3 | sed 's/err [0-9.e+-]*//g'; done
done > $OUT/aux.txt 2>&1
grep -v "^+" $OUT/aux.txt | cut -c1-200
