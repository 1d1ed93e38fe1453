# Round 5: compiler scheduling strategies on wino43.hip (the K loop's instruction interleaving is the compiler's): scratch builds, same-box A/B.
#   bash tests/probe/r05_sched.sh build   (no GPU)   /   bash tests/probe/r05_sched.sh run   (GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
C=$ROOT/v-diffusion-torch_amd/csrc
L=$ROOT/v-diffusion-torch_amd/lib/exp
if [ "$1" = build ]; then
  mkdir -p $L
  for cfg in "maxilp:-mllvm -amdgpu-sched-strategy=max-ilp" "maxmem:-mllvm -amdgpu-sched-strategy=max-memory-clause" "iterilp:-mllvm -amdgpu-sched-strategy=iterative-ilp" "iterminreg:-mllvm -amdgpu-sched-strategy=iterative-minreg" "igrouplp:-mllvm -amdgpu-igrouplp=1"; do
    n=${cfg%%:*}; f=${cfg#*:}
    if /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function $f -c $C/wino43.hip -o /tmp/sched_$n.o 2>/tmp/sched_$n.err; then
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libsched_$n.so $C/gemm.o $C/wino.o /tmp/sched_$n.o $C/attn.o $C/norm.o $C/misc.o $C/diffusion.o $C/optim.o $C/api.o
      echo "built $n"
    else
      echo "FAILED $n: $(tail -2 /tmp/sched_$n.err)"
    fi
  done
else
  OUT=$ROOT/gpurun_out/r05_sched.txt
  : > $OUT
  export VD_PERF_SHAPES=3
  for rep in 1 2; do
    echo "== default (rep $rep)" >> $OUT
    python $ROOT/tests/perf_wino43.py 2>&1 | grep "FORWARD" >> $OUT
    for f in $L/libsched_*.so; do
      echo "== $(basename $f) (rep $rep)" >> $OUT
      VDIFF_HIP_LIB=$f python $ROOT/tests/perf_wino43.py 2>&1 | grep "FORWARD" >> $OUT
    done
  done
  cat $OUT
fi
