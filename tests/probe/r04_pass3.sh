# Round-4 pass 3: K-blocked V / dM layout of the F(4x4,3x3) weight gradient against the plane-major one (scratch library of the
# previous build), its parity tests, where the step's device copies come from, and the reserved-CU cost at 16 / 32
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c}
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "wgrad or gn_ or producer" 2>&1 | tail -15 > $OUT/wgrad_tests.txt
for i in 1 2; do
  echo "== K-blocked (product)"; python tests/perf_wgrad43.py 2>&1 | grep "F(4,3)" | sed 's/.*| F(4,3)/F(4,3)/'
  echo "== plane-major"; VDIFF_HIP_LIB=$L/exp/libvd_planemajor.so python tests/perf_wgrad43.py 2>&1 | grep "F(4,3)" | sed 's/.*| F(4,3)/F(4,3)/'
done > $OUT/wgrad43_ab.txt 2>&1
python tests/probe/find_copies.py > $OUT/copies.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no-sample --no-cpu-baseline --no-secondary --no-extras"
for n in 16 32; do
  echo "== reserve $n"; VD_BENCH_FORCE_REDUCER=1 VD_RESERVE_CUS=$n MASTER_PORT=$((29600 + n)) $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['multi_gpu'])"
done > $OUT/dp_levers2.txt 2>&1
for i in 1 2 3; do
$B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('step fold', j['ms_per_step'])"
VD_GN_FOLD=0 $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('step two-launch', j['ms_per_step'])"
done >> $OUT/dp_levers2.txt 2>&1
python tests/perf_sample.py > $OUT/sample_fold.txt 2>&1; VD_GN_FOLD=0 python tests/perf_sample.py > $OUT/sample_nofold.txt 2>&1
tail -3 $OUT/sample_fold.txt $OUT/sample_nofold.txt
cat $OUT/wgrad_tests.txt $OUT/wgrad43_ab.txt; head -50 $OUT/copies.txt; cat $OUT/dp_levers2.txt
