# Round-4 pass 2: per-shape vd_gemm tables of the CIFAR step under the tile / K-tile knobs (which shapes lose to their natural choice?)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4b}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_multigpu_path_gpu.py -q -m gpu -s -x -k "b128_rows or two_ranks_equal" 2>&1 | tail -40 > $OUT/newtests.txt
python tests/probe/gemm_shapes.py cifar10 128 > $OUT/gemm_natural.txt 2>&1
VD_GEMM_KT=16 python tests/probe/gemm_shapes.py cifar10 128 > $OUT/gemm_kt16.txt 2>&1
VD_GEMM_KT=32 python tests/probe/gemm_shapes.py cifar10 128 > $OUT/gemm_kt32.txt 2>&1
VD_GEMM_TILE=64 python tests/probe/gemm_shapes.py cifar10 128 > $OUT/gemm_t64.txt 2>&1
VD_GEMM_TILE=12864 python tests/probe/gemm_shapes.py cifar10 128 > $OUT/gemm_t12864.txt 2>&1
VD_GEMM_TILE=64128 python tests/probe/gemm_shapes.py cifar10 128 > $OUT/gemm_t64128.txt 2>&1
VD_GEMM_TILE=128 python tests/probe/gemm_shapes.py cifar10 128 > $OUT/gemm_t128.txt 2>&1
python tests/probe/gemm_shapes.py celeba 128 > $OUT/gemm_celeba_natural.txt 2>&1
tail -20 $OUT/newtests.txt
head -50 $OUT/gemm_natural.txt
# 1-GPU cost of the data-parallel levers (1-rank RCCL group, reducer forced on): reserved CUs, readiness per block
B="python bench.py --steps 20 --warmup 5 --no-sample --no-cpu-baseline --no-secondary --no-extras"
for rep in 1 2; do
  for n in 0 8 16 32; do
    echo "== reserve $n"; VD_BENCH_FORCE_REDUCER=1 VD_RESERVE_CUS=$n MASTER_PORT=$((29600 + n)) $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['multi_gpu'])"
  done
  echo "== ready per block"; VD_BENCH_FORCE_REDUCER=1 VD_READY_PER_BLOCK=1 MASTER_PORT=29611 $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['multi_gpu'])"
  echo "== no reducer"; $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'])"
done > $OUT/dp_levers.txt 2>&1
cat $OUT/dp_levers.txt
