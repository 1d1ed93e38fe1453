"""Imported FIRST by the probe scripts that need in-kernel timestamps / timing-only kernel variants: points the binding at
libvdiff_hip_probe.so (csrc built with -DVD_PROBES; `make -C v-diffusion-torch_amd/csrc` builds it beside the product library).
The product library has none of that code, so these scripts cannot run against it."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PROBE_LIB = os.path.join(ROOT, "v-diffusion-torch_amd", "lib", "libvdiff_hip_probe.so")
if not os.path.exists(PROBE_LIB):
    raise SystemExit(f"{PROBE_LIB} missing: make -C v-diffusion-torch_amd/csrc")
os.environ["VDIFF_HIP_LIB"] = PROBE_LIB
