export VDIFF_HIP_LIB=$PWD/v-diffusion-torch_amd/lib/libvdiff_hip_probe.so VD_SOAK_STEPS=10
python tests/soak_gn_split.py 2>/dev/null | grep "^{" | cut -c1-200
VD_SOAK_REDUCER=1 MASTER_PORT=29573 python tests/soak_gn_split.py 2>/dev/null | grep "^{" | cut -c1-200
VD_SOAK_REDUCER=1 VD_RESERVE_CUS=8 MASTER_PORT=29575 python tests/soak_gn_split.py 2>/dev/null | grep "^{" | cut -c1-200
python tests/soak_gn_split.py 2>/dev/null | grep "^{" | cut -c1-200
