# Round 5: start-up skew of the persistent F(4x4,3x3) workgroups (VD_W43_SKEW_*): same-box A/B of scratch builds.
#   bash tests/probe/r05_skew.sh build     here (no GPU): v-diffusion-torch_amd/lib/exp/libskew_<nph>_<sleeps>.so
#   bash tests/probe/r05_skew.sh run       on the GPU box
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
C=$ROOT/v-diffusion-torch_amd/csrc
L=$ROOT/v-diffusion-torch_amd/lib/exp
if [ "$1" = build ]; then
  mkdir -p $L
  for cfg in "4 1 0" "4 2 0" "8 1 0" "2 2 0" "4 4 0" "4 0 1" "4 2 1"; do
    set -- $cfg
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -DVD_W43_SKEW_NPH=$1 -DVD_W43_SKEW_SLEEPS=$2 -DVD_W43_PRIO=$3 -c $C/wino43.hip -o /tmp/skew_$1_$2_$3.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libskew_$1_$2_$3.so $C/gemm.o $C/wino.o /tmp/skew_$1_$2_$3.o $C/attn.o $C/norm.o $C/misc.o $C/diffusion.o $C/optim.o $C/api.o
  done
else
  OUT=$ROOT/gpurun_out/r05_skew.txt
  : > $OUT
  for rep in 1 2; do
    echo "== default (rep $rep)" >> $OUT
    python $ROOT/tests/perf_wino43.py 2>&1 | grep "FORWARD" | sed 's/err [0-9.e+-]*//g' >> $OUT
    for f in $L/libskew_*.so; do
      echo "== $(basename $f) (rep $rep)" >> $OUT
      VDIFF_HIP_LIB=$f python $ROOT/tests/perf_wino43.py 2>&1 | grep "FORWARD" | sed 's/err [0-9.e+-]*//g' >> $OUT
    done
  done
  cat $OUT
fi
