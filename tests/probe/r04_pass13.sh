set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4o
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib/exp
for i in 1 2; do
  echo "== non-temporal (product)"; python tests/perf_gn.py 2>&1 | grep "TB/s"
  echo "== plain"; VDIFF_HIP_LIB=$L/libvd_gn_nt0.so python tests/perf_gn.py 2>&1 | grep "TB/s"
done > $OUT/gn_nt.txt 2>&1
grep -v "^+" $OUT/gn_nt.txt | cut -c1-150
