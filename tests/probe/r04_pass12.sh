# Round-4 pass 12: item order of the F(4x4,3x3) convolution kernel inside an XCD -- 4 channel blocks x 8 tile groups (product) against 8 x 4
# (scratch library): time and FETCH_SIZE / WRITE_SIZE of the same launches
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4l
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib/exp
for i in 1 2; do
  echo "== 4x8 (product)"; python tests/perf_wino43.py 2>&1 | grep "FORWARD" | head -4 | sed 's/err [0-9.e+-]*//g'
  echo "== 8x4"; VDIFF_HIP_LIB=$L/libvd_order1.so python tests/perf_wino43.py 2>&1 | grep "FORWARD" | head -4 | sed 's/err [0-9.e+-]*//g'
done > $OUT/order.txt 2>&1
grep -v "^+" $OUT/order.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f0 -- python3 $GRAFT_REPO_ROOT/tests/probe/wino_pmc_target.py > $OUT/f0.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w0 -- python3 $GRAFT_REPO_ROOT/tests/probe/wino_pmc_target.py > $OUT/w0.log 2>&1
export VDIFF_HIP_LIB=$L/libvd_order1.so
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f1 -- python3 $GRAFT_REPO_ROOT/tests/probe/wino_pmc_target.py > $OUT/f1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w1 -- python3 $GRAFT_REPO_ROOT/tests/probe/wino_pmc_target.py > $OUT/w1.log 2>&1
find $OUT -name "*agent_info.csv" -delete
