# Timing experiments on the F(4x4,3x3) input-gradient kernel (WRONG results by construction).  Two steps:
#   bash tests/probe/w43_exp.sh build     here (no GPU): scratch libraries v-diffusion-torch_amd/lib/exp/libw43_<n>.so with -DVD_W43_EXP=n
#                                          (git-ignored; nothing of this reaches the product or the probe library)
#   bash tests/probe/w43_exp.sh run       on the GPU box: times tests/perf_wino43.py against each
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
C=$ROOT/v-diffusion-torch_amd/csrc
L=$ROOT/v-diffusion-torch_amd/lib/exp
if [ "$1" = build ]; then
  mkdir -p $L
  for e in ${W43_EXPS:-0 1 2 3 4 5}; do
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -DVD_W43_EXP=$e -DVD_W43_SCRATCH_BUILD $W43_EXTRA -c $C/wino43.hip -o /tmp/w43_$e.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libw43_$e.so $C/gemm.o $C/wino.o /tmp/w43_$e.o $C/attn.o $C/norm.o $C/misc.o $C/diffusion.o $C/optim.o $C/api.o
  done
else
  for f in $L/libw43_*.so; do
    echo "== $(basename $f)"
    VDIFF_HIP_LIB=$f python $ROOT/tests/perf_wino43.py 2>&1 | grep "F(4,3)" | sed 's/.*| F(4,3)/F(4,3)/'
  done
fi
