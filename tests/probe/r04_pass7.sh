# Round-4 pass 7: U-fragment prefetch distance of the F(4x4,3x3) kernel (1 = product, 2, 3: scratch libraries)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4g}
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib
for i in 1 2; do
  echo "== UPF 1 (product)"; python tests/perf_wino43.py 2>&1 | grep "FORWARD"
  echo "== UPF 2"; VDIFF_HIP_LIB=$L/exp/libvd_upf2.so python tests/perf_wino43.py 2>&1 | grep "FORWARD"
  echo "== UPF 3"; VDIFF_HIP_LIB=$L/exp/libvd_upf3.so python tests/perf_wino43.py 2>&1 | grep "FORWARD"
done > $OUT/upf_ab.txt 2>&1
grep -v "^+" $OUT/upf_ab.txt | cut -c1-200
