"""MI355X-native v-diffusion hot path behind the call surface of tqch/v-diffusion-torch
(reference v_diffusion/__init__.py:1-21).

The three hot-path names are implemented here.  The other nine names the reference package re-exports (data loading,
config helpers, its Trainer / Evaluator: control plane, out of scope) are NOT re-implemented: when the environment
variable ``VDIFF_REFERENCE_ROOT`` points at a checkout of the reference they are resolved lazily from it (loaded under
the alias ``v_diffusion_ref`` so its relative imports stay inside the reference), which lets the reference's
``train.py`` / ``generate.py`` run unchanged with this package first on ``sys.path``: they receive this package's
``UNet`` / ``GaussianDiffusion`` / ``get_logsnr_schedule`` and the reference's own plumbing around them."""
import importlib.util
import os
import sys

from .diffusion import GaussianDiffusion, get_logsnr_schedule
from .models.unet import UNet

_HOT = ["GaussianDiffusion", "get_logsnr_schedule", "UNet"]
_DELEGATED = ["get_dataloader", "DATA_INFO", "dict2str", "seed_all", "update_config", "fill_with_defaults", "Trainer",
              "Evaluator", "DummyScheduler"]
_ref_pkg = None


def _reference():
    global _ref_pkg
    if _ref_pkg is None:
        root = os.environ.get("VDIFF_REFERENCE_ROOT")
        init = os.path.join(root, "v_diffusion", "__init__.py") if root else None
        if not init or not os.path.exists(init):
            raise ImportError("this name belongs to the reference's control plane, which this package does not re-implement; "
                              "set VDIFF_REFERENCE_ROOT to a tqch/v-diffusion-torch checkout to have it delegated")
        spec = importlib.util.spec_from_file_location("v_diffusion_ref", init,
                                                      submodule_search_locations=[os.path.dirname(init)])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["v_diffusion_ref"] = mod
        try:
            spec.loader.exec_module(mod)
        except BaseException:
            sys.modules.pop("v_diffusion_ref", None)
            raise
        _ref_pkg = mod
    return _ref_pkg


def __getattr__(name):
    if name in _DELEGATED:
        return getattr(_reference(), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


# `from v_diffusion import *` (reference train.py) resolves every name of __all__: list the delegated ones only when they can be
__all__ = _HOT + (_DELEGATED if os.environ.get("VDIFF_REFERENCE_ROOT") else [])
