# Round 5: lead time of the K-tile DMA in wino43_conv_kernel -- which pieces go out a whole tile ahead (behind the previous tile's barrier) and which
# only 8 MFMA steps ahead.  Scratch builds:  bash tests/probe/r05_dma_order.sh build (no GPU) / run (GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
C=$ROOT/v-diffusion-torch_amd/csrc
L=$ROOT/v-diffusion-torch_amd/lib/exp
if [ "$1" = build ]; then
  mkdir -p $L
  for cfg in "pf1_sb12_dps1:-DVD_W43_PATCH_FIRST=1" "pf0_sb13_dps2:-DVD_W43_SB=13 -DVD_W43_DPS=2" "pf1_sb13_dps2:-DVD_W43_PATCH_FIRST=1 -DVD_W43_SB=13 -DVD_W43_DPS=2" "pf1_sb14_dps2:-DVD_W43_PATCH_FIRST=1 -DVD_W43_SB=14 -DVD_W43_DPS=2" "pf0_sb10_dps1:-DVD_W43_SB=10" "pf1_sb10_dps1:-DVD_W43_PATCH_FIRST=1 -DVD_W43_SB=10"; do
    n=${cfg%%:*}; f=${cfg#*:}
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function $f -c $C/wino43.hip -o /tmp/dma_$n.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libdma_$n.so $C/gemm.o $C/wino.o /tmp/dma_$n.o $C/attn.o $C/norm.o $C/misc.o $C/diffusion.o $C/optim.o $C/api.o
  done
else
  OUT=$ROOT/gpurun_out/r05_dma_order.txt
  : > $OUT
  export VD_PERF_SHAPES=4
  for rep in 1 2; do
    echo "== default (rep $rep)" >> $OUT
    python $ROOT/tests/perf_wino43.py 2>&1 | grep "FORWARD" >> $OUT
    for f in $L/libdma_*.so; do
      echo "== $(basename $f) (rep $rep)" >> $OUT
      VDIFF_HIP_LIB=$f python $ROOT/tests/perf_wino43.py 2>&1 | grep "FORWARD" >> $OUT
    done
  done
  cat $OUT
fi
