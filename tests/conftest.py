import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "v-diffusion-torch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The suite runs the PRODUCT DEFAULT dispatch: small batches follow the occupancy rule (vd_conv3x3_wino43_preferred), i.e. the finer
# F(2x2,3x3) items where F(4x4,3x3) would leave CUs idle.  The handful of tests that must push a B <= 64 network through the F(4x4,3x3) kernels of
# the B = 128 step set VD_WINO43_OCC=0 in their own subprocess (test_unet_gpu.py::test_small_batch_networks_through_the_f43_kernels_in_subprocess,
# test_bench_shapes_gpu.py::test_train_steps_through_the_f43_kernels_in_subprocess).
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
